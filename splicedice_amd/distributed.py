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

    def allsum(self, value):
        return int(value)


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

    def allsum(self, value):
        """sum of one integer over the ranks (error counts: every rank must take the same decision)"""
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t[0])


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

    def allgather_into(self, x, recv):
        self.ctx.allgather_dev(x, recv)
        return recv

    def alltoall_into(self, x, recv):
        self.ctx.alltoall_dev(x, recv, x.nbytes // self.world)
        return recv

    def alltoall(self, x):
        """grouped ncclSend/ncclRecv (sdice_alltoall_dev): block q -> rank q over its direct xGMI link"""
        if not _is_dev(x):
            x = self.ctx.to_device(np.ascontiguousarray(x))
            return self.alltoall(x).to_host()
        recv = self.ctx.empty(x.shape, x.dtype)
        self.ctx.alltoall_dev(x, recv, x.nbytes // self.world)
        return recv

    def allsum(self, value):
        return int(self.allgather(np.asarray([int(value)], dtype=np.int64)).sum())


def stat_layout(m):
    """The per-junction table of one rank (m rows, padded) as ONE byte block for ONE all-gather (north_star: "a single
    RCCL all-gather ... to reassemble the output tables"): name -> (byte offset, dtype); every vector starts at a
    multiple of 16 bytes.  -> (offsets, block bytes)"""
    off, at = {}, 0
    for name, dt in zip(STAT_NAMES, _STAT_DTYPES):
        off[name] = (at, np.dtype(dt))
        at += (m * np.dtype(dt).itemsize + 15) // 16 * 16
    return off, at


def pack_stats_host(stats, m):
    off, nbytes = stat_layout(m)
    buf = np.zeros(nbytes, dtype=np.uint8)
    for name, (at, dt) in off.items():
        buf[at: at + m * dt.itemsize] = np.ascontiguousarray(stats[name], dtype=dt).view(np.uint8)
    return buf


def unpack_stats_host(gathered, m, world):
    """[world * block bytes] -> name -> array of world * m entries (rank blocks concatenated)"""
    off, nbytes = stat_layout(m)
    blocks = np.asarray(gathered, dtype=np.uint8).reshape(world, nbytes)
    return {name: np.concatenate([blocks[r, at: at + m * dt.itemsize].view(dt) for r in range(world)])
            for name, (at, dt) in off.items()}


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
    """(host engines: the CPU test double)  PS -> '.3f' quantise -> rank-sum for rows [first, first + k) of one shard
    (with its halo rows) -> dict name -> array of length pad_to (default k), zero beyond k."""
    pad_to = k if pad_to is None else pad_to
    out = {name: np.zeros(pad_to, dt) for name, dt in zip(STAT_NAMES, _STAT_DTYPES)}
    if k:
        ps = engine.quantize3(engine.ps(counts_ext, rp, cl))   # the _allPS.tsv text round trip
        r = engine.ranksum(ps[first: first + k], g1, g2)
        for name in STAT_NAMES:
            out[name][:k] = r[name]
    return out


def _csr_host(sh):
    """(row_ptr, col, nnz) of the rows [ext_lo, ext_hi) of a shard object as the device holds them (after a step)"""
    rp = sh.d_rp.to_host()
    nnz = int(rp[-1])
    if sh.d_j is not None:
        nnz, _ = sh.e.cluster_status()               # (resolves the asynchronous clustering: deferred errors surface here)
    cl = sh.d_cl.offset(0, (nnz,)).to_host() if nnz else np.zeros(0, np.int32)
    return rp, cl, nnz


class CompareShard:
    """One rank's part of quant -> compare_sample_sets, resident in HBM (device engines).

    load() uploads the rank's count rows and EITHER its local CSR (cut out of a replicated clustering by the caller)
    OR the coordinates of its rows [ext_lo, ext_hi) -- then step() clusters that range itself (sdice_cluster_dev,
    asynchronous: no rank ever clusters the whole junction set; the lists of the own rows are complete by the plan's
    construction, shard.shard_plan_junctions).  step() is device work only: [clustering of the range,] PS with the
    '.3f' round trip fused into the store, rank-sum into ONE packed block (stat_layout), ONE all-gather of that
    block, p / tested made contiguous with two strided copies, Benjamini-Hochberg over the gathered vector
    (sdice_bh_masked_dev: padding and untested rows are absent); result() downloads and drops the padding.
    bench.py --workload e2e --gpus N times exactly this step."""

    def __init__(self, engine, comm, n, s, plan, g1, g2):
        self.e, self.comm, self.n, self.s, self.plan = engine, comm, n, s, plan
        part = plan[comm.rank]
        self.lo, self.hi, self.elo = part["own_lo"], part["own_hi"], part["ext_lo"]
        self.k = self.hi - self.lo
        self.m = max(max(p["own_hi"] - p["own_lo"] for p in plan), 1)
        self.off, self.block = stat_layout(self.m)
        from .engine import DeviceArray
        self.packed = engine.empty(self.block, np.uint8).zero()
        self.views = {name: DeviceArray(engine, (self.m,), dt, ptr=self.packed.ptr + at, owned=False)
                      for name, (at, dt) in self.off.items()}
        self.d_g1, self.d_g2 = engine.to_device(g1, np.int32), engine.to_device(g2, np.int32)
        w = comm.world
        self.d_p_all, self.d_t_all = engine.empty(w * self.m, np.float64), engine.empty(w * self.m, np.uint8)
        self.d_q = engine.empty(w * self.m, np.float64)
        self.recv = engine.empty(w * self.block, np.uint8) if w > 1 else None      # (allocated once: step() is malloc-free)
        self.d_counts = self.d_rp = self.d_cl = self.d_ps = self.d_row_of = None
        self.d_j = None

    def load(self, counts_ext, rp=None, cl=None, junctions=None):
        """counts_ext: rows [ext_lo, ext_hi); either (rp, cl) -- the local CSR -- or junctions = (chrom_rank, left,
        right, strand) of those rows in row order"""
        e = self.e
        if self.k:
            self.d_counts = e.to_device(np.ascontiguousarray(counts_ext), np.int32)
            rows = self.d_counts.shape[0]
            if junctions is not None:
                cr, l, r, st = junctions
                assert len(cr) == rows, (len(cr), rows)
                self.d_j = [e.to_device(cr, np.int32), e.to_device(l, np.int32), e.to_device(r, np.int32), e.to_device(st, np.int8)]
                self.d_row_of, self.d_rp = e.empty(rows, np.int32), e.empty(rows + 1, np.int64)
            else:
                self.d_rp = e.to_device(rp, np.int64)
                self.d_cl = e.to_device(cl if cl.size else np.zeros(1, np.int32), np.int32)
            self.d_ps = e.empty(self.d_counts.shape, np.float32)

    def step(self):
        e, k, m, w = self.e, self.k, self.m, self.comm.world
        if k:
            d_cl = self.d_cl
            if self.d_j is not None:                 # the rank's own range, enqueued without a host round trip
                d_cl, _ = e.cluster_dev(*self.d_j, self.d_row_of, self.d_rp, sync=False)
                self.d_cl = d_cl                     # (a view of the context's list buffer: not owned)
            e.set_param("ps.quantize3", 1)          # the '.3f' round trip is fused into the PS store
            try:
                e.ps_dev(self.d_counts, self.d_rp, d_cl, None, self.d_ps)
            finally:
                e.set_param("ps.quantize3", 0)
            first = self.lo - self.elo
            e.ranksum_dev(self.d_ps.offset(first * self.s, (k, self.s)), self.d_g1, self.d_g2,
                          {name: v.offset(0, (k,)) for name, v in self.views.items()})
        if w > 1 and hasattr(self.comm, "allgather_into"):
            self.comm.allgather_into(self.packed, self.recv)          # ONE collective: world x block bytes
        else:
            self.recv = self.comm.allgather(self.packed)
        at_p, at_t = self.off["p"][0], self.off["tested"][0]
        e.copy2d_dev(self.d_p_all.ptr, m * 8, self.recv.ptr + at_p, self.block, m * 8, w)      # rank blocks -> one vector
        e.copy2d_dev(self.d_t_all.ptr, m, self.recv.ptr + at_t, self.block, m, w)
        e.bh_masked_dev(self.d_p_all, self.d_t_all, self.d_q)

    def csr_host(self):
        return _csr_host(self)

    def result(self):
        host = unpack_stats_host(self.recv.to_host(), self.m, self.comm.world)
        host["corrected"] = self.d_q.to_host()
        return host

    def free(self):
        recv = self.recv if self.recv is not self.packed else None
        for a in (self.d_counts, self.d_rp, self.d_cl, self.d_ps, self.d_row_of, self.d_g1, self.d_g2, self.packed, self.d_p_all,
                  self.d_t_all, self.d_q, recv, *(self.d_j or ())):
            if a is not None:
                a.free()


def quant_compare_sharded(engine, comm, counts_ext, row_ptr, col, g1, g2, plan=None, junctions_ext=None):
    """counts_ext: int32 rows [ext_lo, ext_hi) of the count table in output row order -- this rank's
    shard only (shard.shard_plan(row_ptr, col, world)[rank]); CSR over all rows; two column groups.

    junctions_ext = (chrom_rank, left, right, strand) of the rows [ext_lo, ext_hi), in row order: the rank clusters
    ITS range itself and no global CSR exists anywhere (row_ptr = col = None; plan = shard.shard_plan_junctions(...)
    is then required) -- the lists of its own rows are complete by the plan's construction.

    Returns dict(tested, p, z, corrected, med1, med2, mean1, mean2, delta) for ALL n rows, identical
    on every rank, plus plan.  `engine`: the HIP Context (device path) or a host double with
    ps / quantize3 / ranksum / bh (/ cluster).  The per-junction table crosses the ranks as ONE packed block in ONE all-gather.
    """
    if junctions_ext is not None:
        if plan is None:
            raise ValueError("junctions_ext needs the plan it was cut by (shard.shard_plan_junctions)")
        n = plan[-1]["own_hi"]
    else:
        n = row_ptr.size - 1
        plan = plan or shard.shard_plan(row_ptr, col, comm.world)
    part = plan[comm.rank]
    lo, hi, elo = part["own_lo"], part["own_hi"], part["ext_lo"]
    k = hi - lo
    max_rows = max(max(p["own_hi"] - p["own_lo"] for p in plan), 1)
    ext = np.ascontiguousarray(_own_slice(counts_ext, n, part)) if k else counts_ext[:0]
    dev = hasattr(engine, "ps_dev")
    rp = cl = None
    if k and junctions_ext is None:
        rp, cl = shard.local_csr(row_ptr, col, part)
    elif k and not dev:
        row_of, rp, cl = engine.cluster(*junctions_ext)             # the range alone; rows arrive in row order
        if not np.array_equal(row_of, np.arange(len(row_of))):
            raise ValueError("junctions_ext must be in output row order")
    if dev and comm.device:
        sh = CompareShard(engine, comm, n, ext.shape[1] if k else len(g1) + len(g2), plan, g1, g2)
        try:
            sh.load(ext, rp, cl, junctions=junctions_ext if k else None)
            sh.step()
            engine.sync()
            host = sh.result()
        finally:
            sh.free()
    else:
        if dev:
            raise NotImplementedError("a device engine needs a device communicator (RcclComm / SingleComm)")
        stats = shard_stats(engine, ext, rp, cl, lo - elo, k, g1, g2, pad_to=max_rows)
        gathered = comm.allgather(pack_stats_host(stats, max_rows))             # ONE collective
        host = unpack_stats_host(gathered, max_rows, comm.world)
        host["corrected"] = _bh_masked_host(engine, host["p"], host["tested"])
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


CHI2_ZERO_MSG = "The internally computed table of expected frequencies has a zero element"


def _chi2_abort(comm, n_bad, n_tables):
    """scipy.stats.chi2_contingency raises on the first table with an empty row or column and the reference run dies
    with it (pairwise_fisher.py:167-179): every rank takes the same decision from the global count"""
    total = comm.allsum(n_bad)
    if total:
        raise ValueError(f"{CHI2_ZERO_MSG} ({total} of {n_tables} sample-pair tables have an empty row or column)")


def _pairwise_host(engine, comm, ext, rp, cl, a0, k, plan, n, pairs, correction, test="fisher"):
    n_bad = 0
    if k:
        excl = engine.ps(ext, rp, cl, want_excl=True, want_ps=False)
        if test == "chi2":
            p, n_bad = engine.chi2_pairs(ext[a0: a0 + k], excl[a0: a0 + k])
        else:
            p = engine.fisher_pairs(ext[a0: a0 + k], excl[a0: a0 + k])
    else:
        p = np.zeros((0, pairs), dtype=np.float64)
    if test == "chi2":
        _chi2_abort(comm, n_bad, n * pairs)
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


class PairwiseShard:
    """One rank's part of `pairwise`, resident in HBM (device engines): load() uploads the rank's count rows and EITHER
    its local CSR OR the coordinates of its rows [ext_lo, ext_hi) (step() then clusters that range itself, as
    CompareShard does) and allocates every exchange buffer ONCE (their shapes follow from the plan); step() is device
    work only, no allocation, no synchronisation -- exclusion sums, the per-pair test of the rank's rows, and the
    correction: "pairwise" packs the p-value matrix into per-rank column blocks ON THE DEVICE (sdice_copy2d_dev),
    all-to-all, column BH on complete columns, all-to-all back, unpack; "all" all-gathers the raw matrix and ranks it
    redundantly (sdice_bh_masked_dev); "none" needs no exchange.  bench.py --workload pairwise --gpus N times exactly
    this step."""

    def __init__(self, engine, comm, n, s, plan, correction="pairwise", test="fisher", overlap_groups=None):
        self.e, self.comm, self.n, self.s, self.plan = engine, comm, n, s, plan
        self.correction, self.test = correction, test
        part = plan[comm.rank]
        self.lo, self.hi, self.elo = part["own_lo"], part["own_hi"], part["ext_lo"]
        self.k = self.hi - self.lo
        self.pairs = s * (s - 1) // 2
        # the exchange that takes the corrected values home goes in G column groups: group g's all-to-all runs on the
        # context's second stream (sdice_comm_fork) while group g + 1 is being corrected.  Every rank derives the same G.
        ranges = pair_column_ranges(self.pairs, comm.world)
        wmin = min(b - a for a, b in ranges) if self.pairs else 0
        if overlap_groups is None:
            overlap_groups = 1 if comm.world == 1 else min(4, max(1, wmin // 1024))
        self.G = max(1, min(int(overlap_groups), max(wmin, 1))) if hasattr(engine, "comm_fork") else 1
        self.rows_of = [q["own_hi"] - q["own_lo"] for q in plan]
        self.maxk = max(max(self.rows_of), 1)
        self.d_p = engine.empty((max(self.k, 1), max(self.pairs, 1)), np.float64)
        self.d_counts = self.d_rp = self.d_cl = self.d_excl = self.d_row_of = None
        self.d_j = None
        self.d_bad = engine.empty(1, np.int64) if test == "chi2" else None
        self.bufs = {}                                        # exchange buffers, allocated by load()
        self.ms = {}                                          # per-collective times of the last timed_collectives()

    def load(self, counts_ext, rp=None, cl=None, junctions=None):
        e, w = self.e, self.comm.world
        if self.k and self.pairs:
            self.d_counts = e.to_device(np.ascontiguousarray(counts_ext), np.int32)
            rows = self.d_counts.shape[0]
            if junctions is not None:
                cr, l, r, st = junctions
                assert len(cr) == rows, (len(cr), rows)
                self.d_j = [e.to_device(cr, np.int32), e.to_device(l, np.int32), e.to_device(r, np.int32), e.to_device(st, np.int8)]
                self.d_row_of, self.d_rp = e.empty(rows, np.int32), e.empty(rows + 1, np.int64)
            else:
                self.d_rp = e.to_device(rp, np.int64)
                self.d_cl = e.to_device(cl if cl.size else np.zeros(1, np.int32), np.int32)
            self.d_excl = e.empty(self.d_counts.shape, np.int64)
        if self.pairs and self.correction == "pairwise":
            ranges = pair_column_ranges(self.pairs, w)
            maxw = max(max(b - a for a, b in ranges), 1)
            a, b = ranges[self.comm.rank]
            shape = (w, self.maxk, maxw)
            # (G > 1: the way back is laid out [group][rank][row][column of the group], so that a group is one contiguous exchange)
            gw = self._group_width(ranges)
            shape2 = (self.G, w, self.maxk, gw) if self.G > 1 else shape
            self.bufs = dict(send=e.empty(shape, np.float64).zero(), back=e.empty(shape2, np.float64).zero(),
                             mine=e.empty((max(self.n, 1), max(b - a, 1)), np.float64))
            if w > 1 or self.G > 1:
                self.bufs.update(got=e.empty(shape, np.float64), got2=e.empty(shape2, np.float64))
        elif self.pairs and self.correction == "all":
            self.bufs = dict(pad=e.empty((self.maxk, self.pairs), np.float64),
                             d_q=e.empty((w * self.maxk, self.pairs), np.float64))
            if w > 1:
                self.bufs["everything"] = e.empty((w * self.maxk, self.pairs), np.float64)

    def _group_ranges(self, width):
        """the G column groups of a rank that owns `width` pair columns: [(lo, hi)] relative to its first column"""
        return [(g * width // self.G, (g + 1) * width // self.G) for g in range(self.G)]

    def _group_width(self, ranges):
        return max(max(hi - lo for lo, hi in self._group_ranges(b - a)) for a, b in ranges) if self.pairs else 1

    def _alltoall(self, x, into):
        if into is not None and hasattr(self.comm, "alltoall_into"):
            return self.comm.alltoall_into(x, into)
        return self.comm.alltoall(x)

    def step(self):
        e, comm, k, n, s, pairs = self.e, self.comm, self.k, self.n, self.s, self.pairs
        a0, d_p, maxk, rows_of = self.lo - self.elo, self.d_p, self.maxk, self.rows_of
        n_bad = 0
        if k and pairs:
            d_cl = self.d_cl
            if self.d_j is not None:                          # the rank's own range, enqueued without a host round trip
                d_cl, _ = e.cluster_dev(*self.d_j, self.d_row_of, self.d_rp, sync=False)
                self.d_cl = d_cl                              # (a view of the context's list buffer: not owned)
            e.ps_dev(self.d_counts, self.d_rp, d_cl, self.d_excl, None)
            inc, exc = self.d_counts.offset(a0 * s, (k, s)), self.d_excl.offset(a0 * s, (k, s))
            if self.test == "chi2":
                e.chi2_pairs_dev(inc, exc, d_p.offset(0, (k, pairs)), self.d_bad)
                n_bad = int(self.d_bad.to_host()[0])
            else:
                e.fisher_pairs_dev(inc, exc, d_p.offset(0, (k, pairs)))
        if self.test == "chi2":
            _chi2_abort(comm, n_bad, n * pairs)
        if self.correction == "pairwise" and pairs > 0:
            ranges = pair_column_ranges(pairs, comm.world)
            maxw = max(max(b - a for a, b in ranges), 1)
            blk = maxk * maxw * 8
            send, back, mine = self.bufs["send"], self.bufs["back"], self.bufs["mine"]
            for q, (a, b) in enumerate(ranges):               # pack my rows of rank q's columns (device, strided)
                if k and b > a:
                    e.copy2d_dev(send.ptr + q * blk, maxw * 8, d_p.ptr + a * 8, pairs * 8, (b - a) * 8, k)
            got = self._alltoall(send, self.bufs.get("got"))
            a, b = ranges[comm.rank]
            w = b - a
            at = 0
            for r in range(comm.world):                       # rank r's rows of MY columns -> one [n, w] table
                if rows_of[r] and w:
                    e.copy2d_dev(mine.ptr + at * w * 8, w * 8, got.ptr + r * blk, maxw * 8, w * 8, rows_of[r])
                at += rows_of[r]
            if self.G == 1:
                if n and w:
                    e.bh_columns_dev(mine.offset(0, (n, w)))
                at = 0
                for r in range(comm.world):
                    if rows_of[r] and w:
                        e.copy2d_dev(back.ptr + r * blk, maxw * 8, mine.ptr + at * w * 8, w * 8, w * 8, rows_of[r])
                    at += rows_of[r]
                got2 = self._alltoall(back, self.bufs.get("got2"))
                for q, (a, b) in enumerate(ranges):               # corrected values back into my rows
                    if k and b > a:
                        e.copy2d_dev(d_p.ptr + a * 8, pairs * 8, got2.ptr + q * blk, maxw * 8, (b - a) * 8, k)
            else:
                # column groups: correct group g, pack it, send it home on the second stream while group g + 1 is corrected
                gw = self._group_width(ranges)
                blkg = maxk * gw * 8
                got2 = self.bufs["got2"]
                for g, (ga, gb) in enumerate(self._group_ranges(w)):
                    if n and gb > ga:
                        e.bh_columns_pitched_dev(mine.offset(ga, (n, gb - ga)), n, gb - ga, w)
                    at = 0
                    for r in range(comm.world):
                        if rows_of[r] and gb > ga:
                            e.copy2d_dev(back.ptr + (g * comm.world + r) * blkg, gw * 8, mine.ptr + (at * w + ga) * 8, w * 8,
                                         (gb - ga) * 8, rows_of[r])
                        at += rows_of[r]
                    e.comm_fork()                                 # the second stream waits for the group's packing ...
                    view_s = back.offset(g * comm.world * maxk * gw, (comm.world, maxk, gw))
                    view_r = got2.offset(g * comm.world * maxk * gw, (comm.world, maxk, gw))
                    if hasattr(comm, "alltoall_into"):
                        comm.alltoall_into(view_s, view_r)        # ... and carries it while the main stream goes on
                    else:
                        e.copy2d_dev(view_r.ptr, view_s.nbytes, view_s.ptr, view_s.nbytes, view_s.nbytes, 1)
                e.comm_join()
                for q, (a, b) in enumerate(ranges):               # corrected values back into my rows
                    for g, (ga, gb) in enumerate(self._group_ranges(b - a)):
                        if k and gb > ga:
                            e.copy2d_dev(d_p.ptr + (a + ga) * 8, pairs * 8, got2.ptr + (g * comm.world + q) * blkg, gw * 8,
                                         (gb - ga) * 8, k)
        elif self.correction == "all" and pairs > 0:
            pad, d_q = self.bufs["pad"].memset(0xBF), self.bufs["d_q"]     # 0xBFBF... is a negative double: "absent"
            if k:
                e.copy2d_dev(pad.ptr, pairs * 8, d_p.ptr, pairs * 8, pairs * 8, k)
            if "everything" in self.bufs and hasattr(comm, "allgather_into"):
                everything = comm.allgather_into(pad, self.bufs["everything"])
            else:
                everything = comm.allgather(pad)
            e.bh_masked_dev(everything, None, d_q)
            if k:
                e.copy2d_dev(d_p.ptr, pairs * 8, d_q.ptr + comm.rank * maxk * pairs * 8, pairs * 8, pairs * 8, k)

    def timed_collectives(self, reps=3):
        """ms per all-to-all of the step's block shape, timed alone (after a step)"""
        if "send" not in self.bufs:
            return {}
        e, send = self.e, self.bufs["send"]
        self._alltoall(send, self.bufs.get("got"))
        e.sync()
        e.timer_start()
        for _ in range(reps):
            self._alltoall(send, self.bufs.get("got"))
        ms = e.timer_stop() / reps
        return {"alltoall_ms": ms, "alltoall_bytes_per_rank": int(send.nbytes), "alltoalls_per_step": 2}

    def csr_host(self):
        return _csr_host(self)

    def result(self):
        self.e.sync()
        k, pairs = self.k, self.pairs
        return self.d_p.offset(0, (k, pairs)).to_host() if k and pairs else np.zeros((k, pairs), dtype=np.float64)

    def free(self):
        for a in (self.d_counts, self.d_rp, self.d_cl, self.d_excl, self.d_row_of, self.d_p, self.d_bad, *(self.d_j or ()),
                  *self.bufs.values()):
            if a is not None:
                a.free()
        self.bufs = {}


def _pairwise_dev(engine, comm, ext, rp, cl, a0, k, plan, n, pairs, correction, test="fisher", junctions=None, overlap_groups=None):
    sh = PairwiseShard(engine, comm, n, ext.shape[1], plan, correction, test, overlap_groups=overlap_groups)
    try:
        sh.load(ext, rp, cl, junctions=junctions)
        sh.step()
        return sh.result()
    finally:
        sh.free()


def pairwise_sharded(engine, comm, counts_ext, row_ptr, col, correction="pairwise", plan=None, test="fisher",
                     junctions_ext=None, overlap_groups=None):
    """`pairwise` with junction rows sharded over ranks.

    counts_ext: int32 rows [ext_lo, ext_hi) of this rank's shard in row order; CSR over all rows.
    Every rank computes the exclusion sums and the s(s-1)/2 Fisher p-values of ITS rows; the
    correction modes are the reference's (pairwise_fisher.py:182-193): "pairwise" (BH down every pair
    column over all junctions), "all" (one BH over the whole matrix), "none".  test = "chi2": the Yates-corrected
    chi-square of --chi2 (pairwise_fisher.py:133-136) on the same shards; a table with a zero expected frequency anywhere
    aborts the run on every rank, as the reference's chi2_contingency does.
    junctions_ext (with plan = shard.shard_plan_junctions(...), row_ptr = col = None): the rank clusters its own rows
    [ext_lo, ext_hi) itself, as in quant_compare_sharded.  overlap_groups: column groups of the exchange that takes the
    corrected values home (device engines; default: by the width of a rank's column range, 1 on one rank).
    Returns dict(p=[k, pairs] for this rank's own rows, own=(lo, hi), plan=...).
    """
    if correction not in ("pairwise", "all", "none"):
        raise ValueError("correction must be pairwise | all | none")
    if test not in ("fisher", "chi2"):
        raise ValueError("test must be fisher | chi2")
    if junctions_ext is not None:
        if plan is None:
            raise ValueError("junctions_ext needs the plan it was cut by (shard.shard_plan_junctions)")
        n = plan[-1]["own_hi"]
    else:
        n = row_ptr.size - 1
        plan = plan or shard.shard_plan(row_ptr, col, comm.world)
    part = plan[comm.rank]
    lo, hi, elo = part["own_lo"], part["own_hi"], part["ext_lo"]
    k = hi - lo
    ext = np.ascontiguousarray(_own_slice(counts_ext, n, part))
    s = ext.shape[1]
    pairs = s * (s - 1) // 2
    dev = hasattr(engine, "fisher_pairs_dev") and comm.device
    rp, cl = np.zeros(1, np.int64), np.zeros(0, np.int32)
    if k and junctions_ext is None:
        rp, cl = shard.local_csr(row_ptr, col, part)
    elif k and not dev:
        row_of, rp, cl = engine.cluster(*junctions_ext)
        if not np.array_equal(row_of, np.arange(len(row_of))):
            raise ValueError("junctions_ext must be in output row order")
    if dev:
        p = _pairwise_dev(engine, comm, ext, rp, cl, lo - elo, k, plan, n, pairs, correction, test,
                          junctions=junctions_ext if k else None, overlap_groups=overlap_groups)
    else:
        p = _pairwise_host(engine, comm, ext, rp, cl, lo - elo, k, plan, n, pairs, correction, test)
    return dict(p=p, own=(lo, hi), plan=plan)
