"""Thin numpy-facing wrapper over the C ABI (include/sdice.h).  No compute happens here."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import SdiceError, check  # noqa: F401


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class DeviceArray:
    """A device allocation with a numpy-like shape/dtype tag (owned by a Context)."""

    def __init__(self, ctx, shape, dtype, ptr=None, owned=True):
        self.ctx = ctx
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.owned = owned
        if ptr is None:
            p = C.c_void_p()
            check(ctx.lib.sdice_dmalloc(ctx.h, self.nbytes, C.byref(p)), "sdice_dmalloc")
            self.ptr = p.value
        else:
            self.ptr = ptr

    def upload(self, host):
        host = _c(host, self.dtype)
        assert host.nbytes == self.nbytes, (host.nbytes, self.nbytes)
        check(self.ctx.lib.sdice_h2d(self.ctx.h, self.ptr, _ptr(host), self.nbytes), "sdice_h2d")
        return self

    def to_host(self, out=None):
        """-> host copy (into `out`, a C-contiguous array of the same dtype and size, when given)"""
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        else:
            assert out.dtype == self.dtype and out.flags.c_contiguous and out.nbytes == self.nbytes
        check(self.ctx.lib.sdice_d2h(self.ctx.h, _ptr(out), self.ptr, self.nbytes), "sdice_d2h")
        return out

    def memset(self, byte):
        """fill with one byte value (async on the context stream); returns self"""
        check(self.ctx.lib.sdice_dmemset(self.ctx.h, self.ptr, int(byte), self.nbytes), "sdice_dmemset")
        return self

    def zero(self):
        return self.memset(0)

    def offset(self, n_elems_lead, shape):
        """View starting n_elems_lead elements in (not owned)."""
        return DeviceArray(self.ctx, shape, self.dtype, ptr=self.ptr + n_elems_lead * self.dtype.itemsize, owned=False)

    def free(self):
        if self.owned and self.ptr:
            self.ctx.lib.sdice_dfree(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


class Context:
    """One context per GPU / per rank.  Fails loudly when no gfx950 device is usable."""

    def __init__(self, device=0):
        self.lib = _ffi.load()
        h = C.c_void_p()
        check(self.lib.sdice_ctx_create(int(device), C.byref(h)), "sdice_ctx_create")
        self.h = h
        self.device = int(device)

    def close(self):
        if self.h:
            self.lib.sdice_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---------------------------------------------------------------- info / params / timing
    def device_info(self):
        name = C.create_string_buffer(160)
        cus = C.c_int()
        mem = C.c_int64()
        check(self.lib.sdice_device_info(self.h, name, 160, C.byref(cus), C.byref(mem)), "sdice_device_info")
        return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=mem.value)

    def set_param(self, name, value):
        check(self.lib.sdice_set_param(self.h, name.encode(), int(value)), f"sdice_set_param({name})")

    def sync(self):
        check(self.lib.sdice_sync(self.h), "sdice_sync")

    def trim(self):
        """synchronise and release the cached device scratch"""
        check(self.lib.sdice_trim(self.h), "sdice_trim")

    def prof_enable(self, on=True):
        """on: False/0 off, True/1 every kernel, 2 dominant kernels only."""
        check(self.lib.sdice_prof_enable(self.h, int(on)), "sdice_prof_enable")

    def prof_reset(self):
        check(self.lib.sdice_prof_reset(self.h), "sdice_prof_reset")

    def prof_query(self, name):
        n = C.c_int64()
        ms = C.c_double()
        check(self.lib.sdice_prof_query(self.h, name.encode(), C.byref(n), C.byref(ms)), "sdice_prof_query")
        return n.value, ms.value

    def prof_report(self):
        buf = C.create_string_buffer(8192)
        check(self.lib.sdice_prof_report(self.h, buf, 8192), "sdice_prof_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out

    def timer_start(self):
        check(self.lib.sdice_timer_start(self.h), "sdice_timer_start")

    def timer_stop(self):
        ms = C.c_double()
        check(self.lib.sdice_timer_stop(self.h, C.byref(ms)), "sdice_timer_stop")
        return ms.value

    # ---------------------------------------------------------------- device memory
    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceArray(self, host.shape, host.dtype).upload(host)

    # ---------------------------------------------------------------- host entry points
    def cluster(self, chrom_rank, left, right, strand):
        """-> (row_of int32[n], row_ptr int64[n+1], col int32[nnz]); SPLICEDICE.py:230-255,96"""
        cr, l, r = _c(chrom_rank, np.int32), _c(left, np.int32), _c(right, np.int32)
        st = _c(strand, np.int8)
        n = cr.size
        row_of = np.empty(n, dtype=np.int32)
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        nnz = C.c_int64()
        check(self.lib.sdice_cluster(self.h, n, _ptr(cr), _ptr(l), _ptr(r), _ptr(st), _ptr(row_of), _ptr(row_ptr),
                                     C.byref(nnz)), "sdice_cluster")
        col = np.empty(nnz.value, dtype=np.int32)
        check(self.lib.sdice_cluster_col(self.h, _ptr(col), col.size), "sdice_cluster_col")
        return row_of, row_ptr, col

    def ps(self, counts, row_ptr, col, want_excl=False, want_ps=True):
        """-> ps float32[n,s] (and excl int64[n,s]); SPLICEDICE.py:297-310"""
        counts = _c(counts, np.int32)
        n, s = counts.shape
        row_ptr, col = _c(row_ptr, np.int64), _c(col, np.int32)
        ps = np.empty((n, s), dtype=np.float32) if want_ps else None
        excl = np.empty((n, s), dtype=np.int64) if want_excl else None
        check(self.lib.sdice_ps(self.h, n, s, _ptr(counts), _ptr(row_ptr), _ptr(col), _ptr(excl), _ptr(ps)), "sdice_ps")
        if want_ps and want_excl:
            return ps, excl
        return ps if want_ps else excl

    def ps_f64(self, counts, row_ptr, col, n_out=None):
        """-> ps float64[n_out,s] of a float64 count table, sums in list order (counts_to_ps.py:58-70).
        Rows n_out.. of `counts` are sources only."""
        counts = _c(counts, np.float64)
        n_rows, s = counts.shape
        n_out = n_rows if n_out is None else int(n_out)
        row_ptr, col = _c(row_ptr, np.int64), _c(col, np.int32)
        if row_ptr.size != n_out + 1:
            raise ValueError("ps_f64: row_ptr must hold n_out + 1 entries")
        ps = np.empty((n_out, s), dtype=np.float64)
        check(self.lib.sdice_ps_f64(self.h, n_out, n_rows, s, _ptr(counts), _ptr(row_ptr), _ptr(col), _ptr(ps)),
              "sdice_ps_f64")
        return ps

    def excl_f64(self, counts, row_ptr, col):
        """-> float64[n,s]: for every row the sum of the listed rows of a float64 table, one addition per row in list
        order (np.sum(counts[mask], axis=0) of pairwise_fisher.py:158-160 when the lists are in table order)"""
        counts = _c(counts, np.float64)
        n, s = counts.shape
        row_ptr, col = _c(row_ptr, np.int64), _c(col, np.int32)
        if row_ptr.size != n + 1:
            raise ValueError("excl_f64: row_ptr must hold n + 1 entries")
        out = np.empty((n, s), dtype=np.float64)
        check(self.lib.sdice_excl_f64(self.h, n, n, s, _ptr(counts), _ptr(row_ptr), _ptr(col), _ptr(out)), "sdice_excl_f64")
        return out

    def mark_low(self, ps, low_flat_idx):
        ps = _c(ps, np.float32)
        idx = _c(low_flat_idx, np.int64)
        check(self.lib.sdice_mark_low(self.h, ps.size, _ptr(ps), _ptr(idx), idx.size), "sdice_mark_low")
        return ps

    def quantize3(self, ps):
        out = np.array(ps, dtype=np.float32, order="C", copy=True)
        check(self.lib.sdice_quantize3(self.h, out.size, _ptr(out)), "sdice_quantize3")
        return out

    def ranksum(self, ps, g1, g2):
        """compareSampleSets.py:216-232 for every row; un-compacted outputs + tested mask."""
        ps = _c(ps, np.float32)
        n, s = ps.shape
        g1, g2 = _c(g1, np.int32), _c(g2, np.int32)
        out = dict(tested=np.zeros(n, np.uint8), p=np.zeros(n, np.float64), z=np.zeros(n, np.float64),
                   med1=np.zeros(n, np.float32), med2=np.zeros(n, np.float32), mean1=np.zeros(n, np.float32),
                   mean2=np.zeros(n, np.float32), delta=np.zeros(n, np.float32))
        check(self.lib.sdice_ranksum(self.h, n, s, _ptr(ps), _ptr(g1), g1.size, _ptr(g2), g2.size,
                                     _ptr(out["tested"]), _ptr(out["p"]), _ptr(out["z"]), _ptr(out["med1"]),
                                     _ptr(out["med2"]), _ptr(out["mean1"]), _ptr(out["mean2"]), _ptr(out["delta"])),
              "sdice_ranksum")
        return out

    def fisher_pairs(self, incl, excl):
        """pairwise_fisher.py:164-179 -> p float64[n, s(s-1)/2]"""
        incl, excl = _c(incl, np.int32), _c(excl, np.int64)
        n, s = incl.shape
        p = np.empty((n, s * (s - 1) // 2), dtype=np.float64)
        check(self.lib.sdice_fisher_pairs(self.h, n, s, _ptr(incl), _ptr(excl), _ptr(p)), "sdice_fisher_pairs")
        return p

    def chi2_pairs(self, incl, excl):
        """pairwise --chi2 (scipy chi2_contingency per pair) -> (p float64[n, s(s-1)/2], n_bad)"""
        incl, excl = _c(incl, np.int32), _c(excl, np.int64)
        n, s = incl.shape
        p = np.empty((n, s * (s - 1) // 2), dtype=np.float64)
        bad = C.c_int64()
        check(self.lib.sdice_chi2_pairs(self.h, n, s, _ptr(incl), _ptr(excl), _ptr(p), C.byref(bad)), "sdice_chi2_pairs")
        return p, bad.value

    def fisher_tables(self, abcd):
        abcd = _c(abcd, np.int64).reshape(-1, 4)
        p = np.empty(abcd.shape[0], dtype=np.float64)
        check(self.lib.sdice_fisher_tables(self.h, abcd.shape[0], _ptr(abcd), _ptr(p)), "sdice_fisher_tables")
        return p

    def bh(self, p):
        p = _c(p, np.float64)
        q = np.empty_like(p)
        check(self.lib.sdice_bh(self.h, p.size, _ptr(p), _ptr(q)), "sdice_bh")
        return q

    def bh_columns(self, p):
        out = np.array(p, dtype=np.float64, order="C", copy=True)
        n, cols = out.shape
        check(self.lib.sdice_bh_columns(self.h, n, cols, _ptr(out)), "sdice_bh_columns")
        return out

    def sort_unique_u64(self, keys):
        """sorted distinct 64-bit keys (the junction union of quant, SPLICEDICE.py:147-228 + :96)"""
        keys = np.array(keys, dtype=np.uint64, order="C", copy=True)
        n_unique = C.c_int64()
        check(self.lib.sdice_sort_unique_u64(self.h, keys.size, _ptr(keys), C.byref(n_unique)), "sdice_sort_unique_u64")
        return keys[: n_unique.value]

    def rowstats(self, data, idx):
        """per-row np.nanmean / np.nanstd over columns idx, bit-identical to numpy in data's dtype
        (float32 / float64) -> (mean, std, n_nan); findOutliers.py:125-135"""
        data = np.ascontiguousarray(data)
        if data.dtype not in (np.float32, np.float64):
            raise TypeError(f"rowstats: unsupported dtype {data.dtype}")
        idx = _c(idx, np.int32)
        n, s = data.shape
        mean, std = np.zeros(n, data.dtype), np.zeros(n, data.dtype)
        n_nan = np.zeros(n, np.int32)
        check(self.lib.sdice_rowstats(self.h, n, s, _ptr(data), 0 if data.dtype == np.float32 else 1, _ptr(idx), idx.size,
                                      _ptr(mean), _ptr(std), _ptr(n_nan)), "sdice_rowstats")
        return mean, std, n_nan

    def similarity(self, ps, mid, sign):
        """similarity.py:25-47 -> (scores int64[s], counts int64[s]); ps float64 [n, s]"""
        ps, mid, sign = _c(ps, np.float64), _c(mid, np.float64), _c(sign, np.int8)
        n, s = ps.shape
        scores, counts = np.zeros(s, np.int64), np.zeros(s, np.int64)
        check(self.lib.sdice_similarity(self.h, n, s, _ptr(ps), _ptr(mid), _ptr(sign), _ptr(scores), _ptr(counts)),
              "sdice_similarity")
        return scores, counts

    # ---------------------------------------------------------------- device entry points
    def cluster_dev(self, d_chrom, d_left, d_right, d_strand, d_row_of, d_row_ptr, sync=True):
        """-> (DeviceArray col view (ctx-owned), nnz).  sync=False enqueues the whole chain without a host
        round trip (nnz is then None; `cluster_status()` / `sync()` report it and any deferred error)."""
        n = d_chrom.shape[0]
        nnz = C.c_int64()
        check(self.lib.sdice_cluster_dev(self.h, n, d_chrom.ptr, d_left.ptr, d_right.ptr, d_strand.ptr, d_row_of.ptr,
                                         d_row_ptr.ptr, C.byref(nnz) if sync else None), "sdice_cluster_dev")
        p = C.c_void_p()
        check(self.lib.sdice_cluster_col_dev(self.h, C.byref(p), None), "sdice_cluster_col_dev")
        if not sync:
            return DeviceArray(self, (0,), np.int32, ptr=p.value, owned=False), None
        return DeviceArray(self, (nnz.value,), np.int32, ptr=p.value, owned=False), nnz.value

    def cluster_status(self):
        """resolve an asynchronous cluster_dev -> (nnz, reach); raises on a deferred error"""
        nnz, reach = C.c_int64(), C.c_int32()
        check(self.lib.sdice_cluster_status(self.h, C.byref(nnz), C.byref(reach)), "sdice_cluster_status")
        return nnz.value, reach.value

    def ps_dev(self, d_counts, d_row_ptr, d_col, d_excl, d_ps):
        n, s = d_counts.shape
        check(self.lib.sdice_ps_dev(self.h, n, s, d_counts.ptr, d_row_ptr.ptr, d_col.ptr if d_col is not None else None,
                                    d_excl.ptr if d_excl is not None else None,
                                    d_ps.ptr if d_ps is not None else None), "sdice_ps_dev")

    def quantize3_dev(self, d_ps):
        check(self.lib.sdice_quantize3_dev(self.h, int(np.prod(d_ps.shape)), d_ps.ptr), "sdice_quantize3_dev")

    def ranksum_dev(self, d_ps, d_g1, d_g2, out):
        n, s = d_ps.shape
        check(self.lib.sdice_ranksum_dev(self.h, n, s, d_ps.ptr, d_g1.ptr, d_g1.shape[0], d_g2.ptr, d_g2.shape[0],
                                         out["tested"].ptr, out["p"].ptr, out["z"].ptr if out.get("z") else None,
                                         out["med1"].ptr, out["med2"].ptr, out["mean1"].ptr, out["mean2"].ptr,
                                         out["delta"].ptr), "sdice_ranksum_dev")

    def fisher_pairs_dev(self, d_incl, d_excl, d_p):
        n, s = d_incl.shape
        check(self.lib.sdice_fisher_pairs_dev(self.h, n, s, d_incl.ptr, d_excl.ptr, d_p.ptr), "sdice_fisher_pairs_dev")

    def fisher_step_stats(self):
        """(useful, issued) lane-steps of the last Fisher launch made with fisher.count_steps = 1"""
        u, t = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.sdice_fisher_step_stats(self.h, C.byref(u), C.byref(t)), "sdice_fisher_step_stats")
        return int(u.value), int(t.value)

    def chi2_pairs_dev(self, d_incl, d_excl, d_p, d_n_bad):
        n, s = d_incl.shape
        check(self.lib.sdice_chi2_pairs_dev(self.h, n, s, d_incl.ptr, d_excl.ptr, d_p.ptr, d_n_bad.ptr), "sdice_chi2_pairs_dev")

    def bh_columns_dev(self, d_p):
        n, cols = d_p.shape
        check(self.lib.sdice_bh_columns_dev(self.h, n, cols, d_p.ptr), "sdice_bh_columns_dev")

    def bh_columns_pitched_dev(self, d_p, n, cols, pitch):
        """BH down the columns of rows x cols values whose rows are `pitch` elements apart (a column range of a wider table)"""
        check(self.lib.sdice_bh_columns_pitched_dev(self.h, int(n), int(cols), int(pitch), d_p.ptr), "sdice_bh_columns_pitched_dev")

    def comm_fork(self):
        check(self.lib.sdice_comm_fork(self.h), "sdice_comm_fork")

    def comm_join(self):
        check(self.lib.sdice_comm_join(self.h), "sdice_comm_join")

    def bh_masked_dev(self, d_p, d_tested, d_q):
        """BH over the present entries (tested != 0, or p >= 0 when d_tested is None); absent -> 0"""
        check(self.lib.sdice_bh_masked_dev(self.h, int(np.prod(d_p.shape)), d_p.ptr,
                                           d_tested.ptr if d_tested is not None else None, d_q.ptr), "sdice_bh_masked_dev")

    def bh_dev(self, d_p, d_q):
        check(self.lib.sdice_bh_dev(self.h, d_p.shape[0], d_p.ptr, d_q.ptr), "sdice_bh_dev")

    # ---------------------------------------------------------------- multi-GPU
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        check(self.lib.sdice_comm_unique_id(self.h, buf), "sdice_comm_unique_id")
        return buf.raw

    def comm_init(self, uid, rank, world):
        assert len(uid) == 128
        check(self.lib.sdice_comm_init(self.h, C.c_char_p(uid), int(rank), int(world)), "sdice_comm_init")

    def allgather_dev(self, d_send, d_recv):
        check(self.lib.sdice_allgather_dev(self.h, d_send.ptr, d_recv.ptr, d_send.nbytes), "sdice_allgather_dev")

    def alltoall_dev(self, d_send, d_recv, bytes_per_peer):
        check(self.lib.sdice_alltoall_dev(self.h, d_send.ptr, d_recv.ptr, int(bytes_per_peer)), "sdice_alltoall_dev")

    def copy2d_dev(self, dst_ptr, dpitch, src_ptr, spitch, width_bytes, rows):
        check(self.lib.sdice_copy2d_dev(self.h, dst_ptr, int(dpitch), src_ptr, int(spitch), int(width_bytes), int(rows)),
              "sdice_copy2d_dev")
