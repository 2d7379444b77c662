"""Sharded quant -> compare and pairwise pipelines: one process per GPU, junction axis sharded
(splicedice_amd/shard.py).

What a rank holds: the CSR (replicated: 8 B + 4 B x degree per junction) and ONLY the count rows
[ext_lo, ext_hi) of its own shard (its rows plus the read-only halo the plan gives it) -- never the
whole [n, s] table.  With the HIP Context everything stays in HBM: the shard is uploaded once, PS is
produced, quantised and consumed in place, the per-junction statistics are all-gathered as device
buffers (RCCL, padded to equal length), Benjamini-Hochberg runs on the gathered device vectors
(sdice_bh_masked_dev), and only the final per-junction table comes back to the host.

The PS matrix itself never crosses GPUs by default: it stays resident in the HBM of the rank that
owns the rows (at config 5 it is 20 GB; each rank streams its shard to the host over its own PCIe
link).  What has to be reassembled is the per-junction table (tested, p, medians, means), 29 B per
junction, because Benjamini-Hochberg ranks the p-values of ALL tested junctions
(compareSampleSets.py:235).  `gather_ps_dev` is the all-gather of the PS shards that
BASELINE.json's north_star names; bench.py times it next to the no-gather design.

pairwise: every rank computes the Fisher p-values of its rows; the reference's default correction is
BH down every pair COLUMN over all junctions (pairwise_fisher.py:187-191), so the p-value matrix is
transposed across ranks -- blocks packed on the device (sdice_copy2d_dev), RCCL all-to-all, column
BH on complete columns, all-to-all back.  `--multiple_test_correction all` (one BH over the whole
n x pairs matrix, pairwise_fisher.py:182-186) all-gathers the raw matrix (config 4: 32 GB, 26 ms of
xGMI time) and every rank ranks it redundantly, keeping its own rows.

Communicators implement rank, world, allgather(x), alltoall(x) for host arrays (GlooComm: the CPU
tests of the N>1 logic; SingleComm) or device arrays (RcclComm; SingleComm).
"""
import numpy as np

from . import shard

STAT_NAMES = ("tested", "p", "z", "med1", "med2", "mean1", "mean2", "delta")
_STAT_DTYPES = (np.uint8, np.float64, np.float64, np.float32, np.float32, np.float32, np.float32, np.float32)


def _is_dev(x):
    return hasattr(x, "ptr") and hasattr(x, "to_host")


class SingleComm:
    rank, world = 0, 1
    device = True          # passes device arrays through untouched

    def allgather(self, x):
        return x

    def alltoall(self, x):
        return x


class GlooComm:
    """Host-side collectives over an initialised torch.distributed (gloo) process group."""
    device = False

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather(self, x):
        """equal-shaped x on every rank -> concatenation along axis 0 in rank order"""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(x))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.numpy() for o in outs], axis=0)

    def alltoall(self, x):
        """x[q] goes to rank q; returns y with y[r] = the block rank r sent here.
        (gloo has no all_to_all on CPU tensors: one all_gather of the stacked blocks, keep slice `rank`.)"""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(x))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.stack([o.numpy()[self.rank] for o in outs])


class RcclComm:
    """Device collectives through the engine context (RCCL over xGMI)."""
    device = True

    def __init__(self, ctx, rank, world, bcast_bytes):
        self.ctx, self.rank, self.world = ctx, rank, world
        # every rank takes part in the broadcast whatever happens on rank 0 (an all-zero id = "rank 0 failed")
        uid, err = None, None
        if rank == 0:
            try:
                uid = ctx.comm_unique_id()
            except Exception as e:       # noqa: BLE001 -- re-raised below, after the broadcast
                uid, err = bytes(128), e
        uid = bcast_bytes(uid, 128)
        if err is not None:
            raise err
        if not any(uid):
            raise RuntimeError("RCCL unique id could not be created on rank 0")
        ctx.comm_init(uid, rank, world)

    def allgather(self, x):
        if not _is_dev(x):
            x = self.ctx.to_device(np.ascontiguousarray(x))
            return self.allgather(x).to_host()
        recv = self.ctx.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), x.dtype)
        self.ctx.allgather_dev(x, recv)
        return recv

    def alltoall(self, x):
        """grouped ncclSend/ncclRecv (sdice_alltoall_dev): block q -> rank q over its direct xGMI link"""
        if not _is_dev(x):
            x = self.ctx.to_device(np.ascontiguousarray(x))
            return self.alltoall(x).to_host()
        recv = self.ctx.empty(x.shape, x.dtype)
        self.ctx.alltoall_dev(x, recv, x.nbytes // self.world)
        return recv


def _own_slice(counts_ext, n, part):
    """the caller hands over rows [ext_lo, ext_hi); a full [n, s] table (single-process callers) is cut here"""
    elo, ehi = part["ext_lo"], part["ext_hi"]
    if counts_ext.shape[0] == ehi - elo:
        return counts_ext
    if counts_ext.shape[0] == n:
        return counts_ext[elo:ehi]
    raise ValueError(f"expected the {ehi - elo} count rows [{elo}, {ehi}) of this rank's shard, got {counts_ext.shape[0]}")


def _bh_masked_host(engine, p, tested):
    """host twin of sdice_bh_masked_dev for engines without device entry points (the CPU test double)"""
    keep = np.flatnonzero(tested if tested is not None else p >= 0)
    q = np.zeros(p.shape, dtype=np.float64)
    if keep.size:
        q[keep] = engine.bh(np.ascontiguousarray(p[keep]))
    return q


def shard_stats(engine, counts_ext, rp, cl, first, k, g1, g2, pad_to=None):
    """PS -> '.3f' quantise -> rank-sum for rows [first, first + k) of one shard (with its halo rows).
    -> dict name -> array of length pad_to (default k), zero beyond k.  Device engine: device arrays, the
    PS shard is produced, quantised and consumed in HBM; host engine (CPU test double): numpy arrays."""
    pad_to = k if pad_to is None else pad_to
    if not hasattr(engine, "ps_dev"):
        out = {name: np.zeros(pad_to, dt) for name, dt in zip(STAT_NAMES, _STAT_DTYPES)}
        if k:
            ps = engine.quantize3(engine.ps(counts_ext, rp, cl))   # the _allPS.tsv text round trip
            r = engine.ranksum(ps[first: first + k], g1, g2)
            for name in STAT_NAMES:
                out[name][:k] = r[name]
        return out
    out = {}
    for name, dt in zip(STAT_NAMES, _STAT_DTYPES):
        out[name] = engine.empty(max(pad_to, 1), dt).zero()
    if k:
        n_ext, s = counts_ext.shape
        d_counts = engine.to_device(counts_ext, np.int32)
        d_rp = engine.to_device(rp, np.int64)
        d_cl = engine.to_device(cl if cl.size else np.zeros(1, np.int32), np.int32)
        d_ps = engine.empty((n_ext, s), np.float32)
        engine.set_param("ps.quantize3", 1)          # the '.3f' round trip is fused into the PS store
        try:
            engine.ps_dev(d_counts, d_rp, d_cl, None, d_ps)
        finally:
            engine.set_param("ps.quantize3", 0)
        d_g1, d_g2 = engine.to_device(g1, np.int32), engine.to_device(g2, np.int32)
        engine.ranksum_dev(d_ps.offset(first * s, (k, s)), d_g1, d_g2, {name: out[name].offset(0, (k,)) for name in STAT_NAMES})
        engine.sync()
        for a in (d_counts, d_rp, d_cl, d_ps, d_g1, d_g2):
            a.free()
    return out


def quant_compare_sharded(engine, comm, counts_ext, row_ptr, col, g1, g2, plan=None):
    """counts_ext: int32 rows [ext_lo, ext_hi) of the count table in output row order -- this rank's
    shard only (shard.shard_plan(row_ptr, col, world)[rank]); CSR over all rows; two column groups.

    Returns dict(tested, p, z, corrected, med1, med2, mean1, mean2, delta) for ALL n rows, identical
    on every rank, plus plan.  `engine`: the HIP Context (device path) or a host double with
    ps / quantize3 / ranksum / bh.
    """
    n = row_ptr.size - 1
    plan = plan or shard.shard_plan(row_ptr, col, comm.world)
    part = plan[comm.rank]
    lo, hi, elo = part["own_lo"], part["own_hi"], part["ext_lo"]
    k = hi - lo
    max_rows = max(max(p["own_hi"] - p["own_lo"] for p in plan), 1)
    rp, cl = shard.local_csr(row_ptr, col, part) if k else (np.zeros(1, np.int64), np.zeros(0, np.int32))
    ext = np.ascontiguousarray(_own_slice(counts_ext, n, part)) if k else counts_ext[:0]
    stats = shard_stats(engine, ext, rp, cl, lo - elo, k, g1, g2, pad_to=max_rows)
    dev = _is_dev(stats["p"])
    if dev and not comm.device:
        stats = {name: a.to_host() for name, a in stats.items()}
        dev = False
    gathered = {name: comm.allgather(stats[name]) for name in STAT_NAMES}       # [world * max_rows] each
    if dev:
        d_q = engine.empty(comm.world * max_rows, np.float64)
        engine.bh_masked_dev(gathered["p"], gathered["tested"], d_q)
        host = {name: gathered[name].to_host() for name in STAT_NAMES}
        host["corrected"] = d_q.to_host()
    else:
        host = dict(gathered)
        host["corrected"] = _bh_masked_host(engine, gathered["p"], gathered["tested"])
    out = {}
    for name, a in host.items():                                                # drop the padding: rows in global order
        out[name] = np.concatenate([a[r * max_rows: r * max_rows + plan[r]["own_hi"] - plan[r]["own_lo"]]
                                    for r in range(comm.world)])
        assert out[name].shape[0] == n
    out["plan"] = plan
    return out


def gather_ps_dev(engine, comm, d_ps_own, k, s, max_rows):
    """north_star's collective: all-gather of the PS shards (padded to max_rows rows) -> device array
    [world * max_rows, s] float32.  Timed by bench.py beside the design that leaves PS where it is."""
    if k == max_rows:
        send = d_ps_own
    else:
        send = engine.empty((max_rows, s), np.float32).zero()
        engine.copy2d_dev(send.ptr, s * 4, d_ps_own.ptr, s * 4, s * 4, k)
    return comm.allgather(send)


def pair_column_ranges(pairs, world):
    """Pair columns owned by each rank for the column-wise BH: contiguous, near-equal."""
    return [(q * pairs // world, (q + 1) * pairs // world) for q in range(world)]


def _pairwise_host(engine, comm, ext, rp, cl, a0, k, plan, n, pairs, correction):
    if k:
        excl = engine.ps(ext, rp, cl, want_excl=True, want_ps=False)
        p = engine.fisher_pairs(ext[a0: a0 + k], excl[a0: a0 + k])
    else:
        p = np.zeros((0, pairs), dtype=np.float64)
    rows_of = [q["own_hi"] - q["own_lo"] for q in plan]
    maxk = max(max(rows_of), 1)
    if correction == "pairwise" and pairs > 0:
        ranges = pair_column_ranges(pairs, comm.world)
        maxw = max(max(b - a for a, b in ranges), 1)
        send = np.zeros((comm.world, maxk, maxw), dtype=np.float64)
        for q, (a, b) in enumerate(ranges):                   # my rows of rank q's columns
            send[q, :k, : b - a] = p[:, a:b]
        got = comm.alltoall(send)                             # rank r's rows of MY columns
        a, b = ranges[comm.rank]
        mine = np.concatenate([got[r][: rows_of[r], : b - a] for r in range(comm.world)], axis=0)
        assert mine.shape == (n, b - a)
        if mine.size:
            mine = engine.bh_columns(mine)
        back, at = np.zeros((comm.world, maxk, maxw), dtype=np.float64), 0
        for r in range(comm.world):                           # corrected values go home
            back[r, : rows_of[r], : b - a] = mine[at: at + rows_of[r]]
            at += rows_of[r]
        got = comm.alltoall(back)                             # my rows of rank q's columns, corrected
        for q, (a, b) in enumerate(ranges):
            p[:, a:b] = got[q][:k, : b - a]
    elif correction == "all" and pairs > 0:
        pad = np.full((maxk, pairs), -1.0)
        pad[:k] = p
        everything = comm.allgather(pad)                      # [world * maxk, pairs], absent rows negative
        q = _bh_masked_host(engine, everything.reshape(-1), None).reshape(everything.shape)
        p = q[comm.rank * maxk: comm.rank * maxk + k]
    return p


def _pairwise_dev(engine, comm, ext, rp, cl, a0, k, plan, n, pairs, correction):
    s = ext.shape[1]
    rows_of = [q["own_hi"] - q["own_lo"] for q in plan]
    maxk = max(max(rows_of), 1)
    d_p = engine.empty((max(k, 1), max(pairs, 1)), np.float64)
    if k and pairs:
        d_counts = engine.to_device(ext, np.int32)
        d_rp = engine.to_device(rp, np.int64)
        d_cl = engine.to_device(cl if cl.size else np.zeros(1, np.int32), np.int32)
        d_excl = engine.empty(ext.shape, np.int64)
        engine.ps_dev(d_counts, d_rp, d_cl, d_excl, None)
        engine.fisher_pairs_dev(d_counts.offset(a0 * s, (k, s)), d_excl.offset(a0 * s, (k, s)), d_p.offset(0, (k, pairs)))
        engine.sync()
        for a in (d_counts, d_rp, d_cl, d_excl):
            a.free()
    if correction == "pairwise" and pairs > 0:
        ranges = pair_column_ranges(pairs, comm.world)
        maxw = max(max(b - a for a, b in ranges), 1)
        blk = maxk * maxw * 8
        send = engine.empty((comm.world, maxk, maxw), np.float64).zero()
        for q, (a, b) in enumerate(ranges):                   # pack my rows of rank q's columns (device, strided)
            if k and b > a:
                engine.copy2d_dev(send.ptr + q * blk, maxw * 8, d_p.ptr + a * 8, pairs * 8, (b - a) * 8, k)
        got = comm.alltoall(send)
        a, b = ranges[comm.rank]
        w = b - a
        mine = engine.empty((max(n, 1), max(w, 1)), np.float64)
        at = 0
        for r in range(comm.world):                           # rank r's rows of MY columns -> one [n, w] table
            if rows_of[r] and w:
                engine.copy2d_dev(mine.ptr + at * w * 8, w * 8, got.ptr + r * blk, maxw * 8, w * 8, rows_of[r])
            at += rows_of[r]
        if n and w:
            engine.bh_columns_dev(mine.offset(0, (n, w)))
        back, at = engine.empty((comm.world, maxk, maxw), np.float64).zero(), 0
        for r in range(comm.world):
            if rows_of[r] and w:
                engine.copy2d_dev(back.ptr + r * blk, maxw * 8, mine.ptr + at * w * 8, w * 8, w * 8, rows_of[r])
            at += rows_of[r]
        got = comm.alltoall(back)
        for q, (a, b) in enumerate(ranges):                   # corrected values back into my rows
            if k and b > a:
                engine.copy2d_dev(d_p.ptr + a * 8, pairs * 8, got.ptr + q * blk, maxw * 8, (b - a) * 8, k)
    elif correction == "all" and pairs > 0:
        pad = engine.empty((maxk, pairs), np.float64).memset(0xBF)     # 0xBFBF... is a negative double: "absent"
        if k:
            engine.copy2d_dev(pad.ptr, pairs * 8, d_p.ptr, pairs * 8, pairs * 8, k)
        everything = comm.allgather(pad)
        d_q = engine.empty(everything.shape, np.float64)
        engine.bh_masked_dev(everything, None, d_q)
        if k:
            engine.copy2d_dev(d_p.ptr, pairs * 8, d_q.ptr + comm.rank * maxk * pairs * 8, pairs * 8, pairs * 8, k)
    engine.sync()
    return d_p.offset(0, (k, pairs)).to_host() if k and pairs else np.zeros((k, pairs), dtype=np.float64)


def pairwise_sharded(engine, comm, counts_ext, row_ptr, col, correction="pairwise", plan=None):
    """`pairwise` with junction rows sharded over ranks.

    counts_ext: int32 rows [ext_lo, ext_hi) of this rank's shard in row order; CSR over all rows.
    Every rank computes the exclusion sums and the s(s-1)/2 Fisher p-values of ITS rows; the
    correction modes are the reference's (pairwise_fisher.py:182-193): "pairwise" (BH down every pair
    column over all junctions), "all" (one BH over the whole matrix), "none".
    Returns dict(p=[k, pairs] for this rank's own rows, own=(lo, hi), plan=...).
    """
    if correction not in ("pairwise", "all", "none"):
        raise ValueError("correction must be pairwise | all | none")
    n = row_ptr.size - 1
    plan = plan or shard.shard_plan(row_ptr, col, comm.world)
    part = plan[comm.rank]
    lo, hi, elo = part["own_lo"], part["own_hi"], part["ext_lo"]
    k = hi - lo
    ext = np.ascontiguousarray(_own_slice(counts_ext, n, part))
    s = ext.shape[1]
    pairs = s * (s - 1) // 2
    rp, cl = shard.local_csr(row_ptr, col, part) if k else (np.zeros(1, np.int64), np.zeros(0, np.int32))
    dev = hasattr(engine, "fisher_pairs_dev") and comm.device
    fn = _pairwise_dev if dev else _pairwise_host
    p = fn(engine, comm, ext, rp, cl, lo - elo, k, plan, n, pairs, correction)
    return dict(p=p, own=(lo, hi), plan=plan)
