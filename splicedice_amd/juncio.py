"""Junction-file ingestion for `quant` on top of the library's multithreaded parser
(sdice_junc_open / sdice_junc_read / sdice_junc_lookup; csrc/juncio.cpp).

Same semantics as the reference's two Python passes over every sample file
(SPLICEDICE.getAllJunctions :147-228, SPLICEDICE.getJunctionCounts :257-295); each file is read
once.  Everything here is host side.
"""
import ctypes as C

import numpy as np

from . import _ffi, _stages

TYPE_CODE = {"bed": 0, "leafcutter": 0, "splicedicebed": 1, "SJ": 2}   # other types contribute nothing (:23-36)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def parse_sample(path, type_code, args, threads=0):
    """-> dict(chroms list[str], chrom_id i32[n], left i32[n], right i32[n], strand i8[n] (0 '+', 1 '-', 2 other),
    score i64[n], admit u8[n])"""
    lib = _ffi.load()
    h = C.c_void_p()
    n, nc, cb = C.c_int64(), C.c_int32(), C.c_int64()
    _ffi.check(lib.sdice_junc_open(str(path).encode(), type_code, C.byref(h), C.byref(n), C.byref(nc), C.byref(cb)),
               "sdice_junc_open")
    try:
        m = n.value
        out = dict(chrom_id=np.empty(m, np.int32), left=np.empty(m, np.int32), right=np.empty(m, np.int32),
                   strand=np.empty(m, np.int8), score=np.empty(m, np.int64), admit=np.empty(m, np.uint8))
        names = C.create_string_buffer(max(1, cb.value))
        off = np.zeros(nc.value + 1, dtype=np.int64)
        rc = lib.sdice_junc_read(h, int(args.minLength), int(args.maxLength), int(args.minUnique), int(args.minOverhang),
                                 float(args.minEntropy), 1 if args.noMultimap else 0, _vp(out["chrom_id"]),
                                 _vp(out["left"]), _vp(out["right"]), _vp(out["strand"]), _vp(out["score"]),
                                 _vp(out["admit"]), names, _vp(off), int(threads))
        if rc != 0:
            raise ValueError(f"{path}: {lib.sdice_last_error().decode()}")
    finally:
        lib.sdice_junc_close(h)
    raw = names.raw[:cb.value]
    out["chroms"] = [raw[off[i]:off[i + 1]].decode() for i in range(nc.value)]
    return out


def lookup_rows(rows, q_chrom, q_left, q_right, q_strand):
    """rows: (chrom_rank, left, right, strand) arrays sorted in row order -> int32 row per query (-1 absent)"""
    lib = _ffi.load()
    rc_, rl, rr, rs = (np.ascontiguousarray(a, dtype=d) for a, d in zip(rows, (np.int32, np.int32, np.int32, np.int8)))
    qc, ql, qr = (np.ascontiguousarray(a, dtype=np.int32) for a in (q_chrom, q_left, q_right))
    qs = np.ascontiguousarray(q_strand, dtype=np.int8)
    out = np.empty(qc.size, dtype=np.int32)
    _ffi.check(lib.sdice_junc_lookup(rc_.size, _vp(rc_), _vp(rl), _vp(rr), _vp(rs), qc.size, _vp(qc), _vp(ql), _vp(qr),
                                     _vp(qs), _vp(out), 0), "sdice_junc_lookup")
    return out


# order-preserving 64-bit packing of (chrom_rank, left, right, strand): 12 | 31 | 20 | 1 bits
_SPAN_BITS, _LEFT_BITS, _CHROM_BITS = 20, 31, 12


def _union_packed(ctx, recs):
    """Sorted junction union through sdice_sort_unique_u64, or None when a junction does not fit the
    packing (more than 4096 chromosomes, span >= 2^20, negative coordinates).  recs carry "keys" / "packable" from
    _rank_and_pack."""
    if not all(rec["packable"] for rec in recs):
        return None
    parts = [rec["keys"] for rec in recs if rec["keys"].size]
    if not parts:
        return tuple(np.zeros(0, d) for d in (np.int32, np.int32, np.int32, np.int8))
    return _unpack_keys(ctx.sort_unique_u64(np.concatenate(parts)))


def _rank_and_pack(rec, rank):
    """rec["chrom_rank"] (the file's chromosome table mapped to the ranks of the sorted union of names) and the admitted
    lines' packed keys: one library call per file, the calls side by side (sdice_junc_pack_keys)."""
    lib = _ffi.load()
    local = np.fromiter((rank[c] for c in rec["chroms"]), dtype=np.int32, count=len(rec["chroms"]))
    n = rec["chrom_id"].size
    rec["chrom_rank"] = np.empty(n, np.int32)
    keys = np.empty(n, np.uint64)
    nk, ok = C.c_int64(), C.c_int32()
    _ffi.check(lib.sdice_junc_pack_keys(n, _vp(rec["chrom_id"]), _vp(local), local.size, _vp(rec["left"]), _vp(rec["right"]),
                                        _vp(rec["strand"]), _vp(rec["admit"]), _vp(rec["chrom_rank"]), _vp(keys), C.byref(nk),
                                        C.byref(ok)), "sdice_junc_pack_keys")
    rec["keys"], rec["packable"] = keys[:nk.value], bool(ok.value)


def _unpack_keys(keys):
    """packed uint64 keys -> (chrom_rank, left, right, strand).  Decoded as UNSIGNED words: chromosome
    ranks >= 2048 set bit 63, which an int64 arithmetic shift would smear into a negative rank."""
    keys = np.asarray(keys, dtype=np.uint64)
    u = np.uint64
    left = (keys >> u(_SPAN_BITS + 1)) & u((1 << _LEFT_BITS) - 1)
    span = (keys >> u(1)) & u((1 << _SPAN_BITS) - 1)
    return ((keys >> u(_LEFT_BITS + _SPAN_BITS + 1)).astype(np.int32), left.astype(np.int32),
            (left + span).astype(np.int32), (keys & u(1)).astype(np.int8))


def ingest(manifest, args, ctx=None):
    """All sample files of a manifest -> (chrom_names_sorted, junction arrays in row order
    (chrom_rank, left, right, strand), parsed per-sample records with global chromosome ranks).
    With an engine context the union / sort of the junction set runs on the GPU."""
    # the files are parsed concurrently (the library call releases the GIL); results keep manifest order
    from concurrent.futures import ThreadPoolExecutor
    cores = host_threads()
    workers = max(1, min(len(manifest), cores))
    per_file = max(1, cores // workers)

    def _one(sample):
        code = TYPE_CODE.get(sample.type)
        return parse_sample(sample.filename, code, args, per_file) if code is not None else None

    with _stages.stage("parse:files"):
        if workers > 1:
            with ThreadPoolExecutor(workers) as pool:
                parsed = list(pool.map(_one, manifest))
        else:
            parsed = [_one(sample) for sample in manifest]
    all_names = set()
    for rec in parsed:
        if rec is not None:
            all_names.update(rec["chroms"])
    names = sorted(all_names)                     # Python string order, as the reference's tuple sorts
    rank = {c: i for i, c in enumerate(names)}
    keys = []
    present = [rec for rec in parsed if rec is not None]
    with _stages.stage("parse:keys"):
        if workers > 1:
            with ThreadPoolExecutor(workers) as pool:
                list(pool.map(lambda rec: _rank_and_pack(rec, rank), present))
        else:
            for rec in present:
                _rank_and_pack(rec, rank)
    if ctx is not None:
        with _stages.stage("parse:union"):
            junc = _union_packed(ctx, present)
        if junc is not None:
            for rec in present:
                del rec["keys"]
            return names, junc, parsed
    for rec in parsed:
        if rec is None:
            continue
        a = rec["admit"].astype(bool)
        if a.any():
            k1 = (rec["chrom_rank"][a].astype(np.int64) << 32) | rec["left"][a].astype(np.int64)
            k2 = (rec["right"][a].astype(np.int64) << 1) | rec["strand"][a].astype(np.int64)
            o = np.lexsort((k2, k1))
            k1, k2 = k1[o], k2[o]
            first = np.r_[True, (k1[1:] != k1[:-1]) | (k2[1:] != k2[:-1])]
            keys.append((k1[first], k2[first]))
    if keys:
        k1 = np.concatenate([k[0] for k in keys])
        k2 = np.concatenate([k[1] for k in keys])
        o = np.lexsort((k2, k1))
        k1, k2 = k1[o], k2[o]
        first = np.r_[True, (k1[1:] != k1[:-1]) | (k2[1:] != k2[:-1])]
        k1, k2 = k1[first], k2[first]
    else:
        k1 = k2 = np.zeros(0, np.int64)
    if k1.size and (k1 < 0).any():
        raise ValueError("junction coordinates must be non-negative")
    junc = ((k1 >> 32).astype(np.int32), (k1 & 0xFFFFFFFF).astype(np.int32), (k2 >> 1).astype(np.int32),
            (k2 & 1).astype(np.int8))
    return names, junc, parsed


def host_threads():
    """worker threads for host-side work: hardware threads capped by the cgroup's CPU quota (the library's own default)"""
    return max(1, int(_ffi.load().sdice_host_threads()))


def gather_counts(manifest, parsed, rows, args):
    """counts int32 [N, S] + flat `low` indices; later lines overwrite earlier ones, no score filter
    (SPLICEDICE.py:257-295).  Every sample fills its own contiguous stretch of the TRANSPOSED table (one library call per
    sample, the calls side by side: look-up and store in one pass, sdice_junc_count_column); one threaded transpose
    gives the [junction][sample] table."""
    from concurrent.futures import ThreadPoolExecutor
    lib = _ffi.load()
    n, s = rows[0].size, len(manifest)
    rc_, rl, rr, rs = (np.ascontiguousarray(a, dtype=d) for a, d in zip(rows, (np.int32, np.int32, np.int32, np.int8)))
    counts_t = np.zeros((s, n), dtype=np.int32)
    want_low = [bool(args.lowCoverageNan and sample.type != "SJ") for sample in manifest]
    low_t = {si: np.zeros(n, dtype=np.uint8) for si in range(s) if want_low[si]}

    def _one(si):
        rec = parsed[si]
        if rec is None or rec["left"].size == 0:
            return
        qc, ql, qr = (np.ascontiguousarray(rec[k], dtype=np.int32) for k in ("chrom_rank", "left", "right"))
        qs = np.ascontiguousarray(rec["strand"], dtype=np.int8)
        sc = np.ascontiguousarray(rec["score"], dtype=np.int64)
        low = low_t.get(si)
        rc = lib.sdice_junc_count_column(n, _vp(rc_), _vp(rl), _vp(rr), _vp(rs), qc.size, _vp(qc), _vp(ql), _vp(qr), _vp(qs),
                                         _vp(sc), int(args.minUnique), _vp(counts_t[si]), _vp(low) if low is not None else None)
        if rc != 0:
            raise ValueError(lib.sdice_last_error().decode())

    with _stages.stage("parse:gather"):
        workers = max(1, min(s, host_threads()))
        if workers > 1:
            with ThreadPoolExecutor(workers) as pool:
                list(pool.map(_one, range(s)))
        else:
            for si in range(s):
                _one(si)
        counts = np.empty((n, s), dtype=np.int32)
        _ffi.check(lib.sdice_transpose_i32(s, n, _vp(counts_t), _vp(counts), 0), "sdice_transpose_i32")
    low = [np.flatnonzero(low_t[si]).astype(np.int64) * s + si for si in sorted(low_t)]
    low = [x for x in low if x.size]
    return counts, (np.concatenate(low) if low else np.zeros(0, np.int64))
