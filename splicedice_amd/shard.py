"""Junction-axis sharding for multi-GPU runs (new; the reference is single-process).

Rows (junctions in output order) are cut into `world` contiguous ranges.  A cut between rows
r-1 and r is *clean* when no overlap edge crosses it; overlap clusters are gene sized, so a
clean cut almost always exists within a few rows of the ideal position k*n/world and the
shard then needs no halo at all.  When none is close enough the shard is extended by the
rows its owned rows reference (read-only halo); outputs of halo rows are discarded.
"""
import numpy as np


def clean_cuts(row_ptr, col):
    """bool[n+1]: cut[r] is True when no CSR edge joins a row < r with a row >= r."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    n = row_ptr.size - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(row_ptr))
    hi = np.arange(n, dtype=np.int64)
    lo = np.arange(n, dtype=np.int64)
    if col.size:
        np.maximum.at(hi, rows, col)
        np.minimum.at(lo, rows, col)
    cut = np.ones(n + 1, dtype=bool)
    if n:
        pm = np.maximum.accumulate(hi)                  # furthest row referenced by rows <= r
        sm = np.minimum.accumulate(lo[::-1])[::-1]      # nearest row referenced by rows >= r
        cut[1:n] = (pm[:-1] < np.arange(1, n)) & (sm[1:] >= np.arange(1, n))
    return cut


def shard_plan(row_ptr, col, world, max_shift_frac=0.02):
    """-> list of dicts(own_lo, own_hi, ext_lo, ext_hi), one per rank (sdice_shard_plan, host C++).

    own_*: rows whose results the rank produces; ext_*: rows it must hold (own + halo).
    """
    import ctypes as C
    from . import _ffi
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = row_ptr.size - 1
    out = np.zeros((world, 4), dtype=np.int64)
    vp = C.c_void_p
    _ffi.check(_ffi.load().sdice_shard_plan(n, row_ptr.ctypes.data_as(vp), col.ctypes.data_as(vp) if col.size else None,
                                            int(world), float(max_shift_frac), out.ctypes.data_as(vp)), "sdice_shard_plan")
    return [dict(own_lo=int(a), own_hi=int(b), ext_lo=int(c), ext_hi=int(d)) for a, b, c, d in out]


def junction_order(chrom_rank, left, right, strand):
    """permutation that puts junctions into output row order (chrom, left, right, strand) -- SPLICEDICE.py:96"""
    return np.lexsort((strand, right, left, chrom_rank))


def shard_plan_junctions(chrom_rank, left, right, strand, world, max_shift_frac=0.02):
    """shard_plan from the coordinates of the junctions IN OUTPUT ROW ORDER (sorted, distinct) -- no CSR, so no rank
    clusters the whole set: a rank clusters its rows [ext_lo, ext_hi) alone and gets the lists of its own rows complete
    (sdice_shard_plan_junctions, host C++)."""
    import ctypes as C
    from . import _ffi
    cr, l, r = (np.ascontiguousarray(x, dtype=np.int32) for x in (chrom_rank, left, right))
    st = np.ascontiguousarray(strand, dtype=np.int8)
    n = cr.size
    out = np.zeros((world, 4), dtype=np.int64)
    vp = C.c_void_p
    _ffi.check(_ffi.load().sdice_shard_plan_junctions(n, cr.ctypes.data_as(vp), l.ctypes.data_as(vp), r.ctypes.data_as(vp),
                                                      st.ctypes.data_as(vp), int(world), float(max_shift_frac),
                                                      out.ctypes.data_as(vp)), "sdice_shard_plan_junctions")
    return [dict(own_lo=int(a), own_hi=int(b), ext_lo=int(c), ext_hi=int(d)) for a, b, c, d in out]


def local_csr(row_ptr, col, part):
    """CSR of the rows [ext_lo, ext_hi) with neighbour indices relative to ext_lo.

    Owned rows keep every neighbour (all inside the extended range by construction); halo rows
    drop neighbours outside it -- their results are never used.
    """
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    lo, hi = part["ext_lo"], part["ext_hi"]
    seg = col[row_ptr[lo]:row_ptr[hi]]
    rows = np.repeat(np.arange(hi - lo, dtype=np.int64), np.diff(row_ptr[lo:hi + 1]))
    keep = (seg >= lo) & (seg < hi)
    deg = np.bincount(rows[keep], minlength=hi - lo)
    rp = np.zeros(hi - lo + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    return rp, (seg[keep] - lo).astype(np.int32)
