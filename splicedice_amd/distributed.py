"""Sharded quant -> compare pipeline: one process per GPU, junction axis sharded
(splicedice_amd/shard.py), one all-gather of the per-junction result table, BH on the
gathered p-values (every rank computes the same corrected vector).

The PS matrix itself never crosses GPUs: it stays resident in the HBM of the rank that owns
the rows (at config 5 it is 20 GB; each rank would stream its shard to the host over its own
PCIe link).  What has to be reassembled is the per-junction table (tested, p, medians, means),
29 B per junction, because Benjamini-Hochberg ranks the p-values of ALL tested junctions
(compareSampleSets.py:235).

Two communicators implement the same three calls (rank, world, allgather_rows):
  RcclComm -- the library's RCCL all-gather on device buffers (sdice_allgather_dev); the
              128-byte unique id travels over any byte channel the caller supplies.
  GlooComm -- torch.distributed gloo on host arrays; used by the CPU tests of the N>1 logic.
"""
import numpy as np

from . import shard

STAT_COLS = 8      # tested, p, z, med1, med2, mean1, mean2, delta  (float64 columns of the gathered table)


class SingleComm:
    rank, world = 0, 1

    def allgather_rows(self, table):
        return [table]

    def alltoall(self, blocks):
        return [blocks[0]]


class GlooComm:
    """Host-side all-gather over an initialised torch.distributed (gloo) process group."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_rows(self, table):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(table))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.numpy() for o in outs]

    def alltoall(self, blocks):
        """blocks[q] (equal shapes) goes to rank q; returns [block from rank r for r in ranks].
        (gloo has no all_to_all on CPU tensors: one all_gather of the stacked blocks, keep column `rank`.)"""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(np.stack(blocks)))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.numpy()[self.rank] for o in outs]


class RcclComm:
    """Device all-gather through the engine context (RCCL over xGMI)."""

    def __init__(self, ctx, rank, world, bcast_bytes):
        self.ctx, self.rank, self.world = ctx, rank, world
        uid = ctx.comm_unique_id() if rank == 0 else None
        ctx.comm_init(bcast_bytes(uid, 128), rank, world)

    def allgather_rows(self, table):
        table = np.ascontiguousarray(table)
        send = self.ctx.to_device(table)
        recv = self.ctx.empty((self.world,) + table.shape, table.dtype)
        self.ctx.allgather_dev(send, recv)
        self.ctx.sync()
        out = recv.to_host()
        return [out[r] for r in range(self.world)]

    def alltoall(self, blocks):
        """RCCL grouped send/recv (sdice_alltoall_dev): blocks[q] -> rank q over its direct xGMI link."""
        stack = np.ascontiguousarray(np.stack(blocks))
        send = self.ctx.to_device(stack)
        recv = self.ctx.empty(stack.shape, stack.dtype)
        self.ctx.alltoall_dev(send, recv, stack[0].nbytes)
        self.ctx.sync()
        out = recv.to_host()
        return [out[r] for r in range(self.world)]


STAT_NAMES = ("tested", "p", "z", "med1", "med2", "mean1", "mean2", "delta")
_STAT_DTYPES = (np.uint8, np.float64, np.float64, np.float32, np.float32, np.float32, np.float32, np.float32)


def _shard_stats(engine, counts_ext, rp, cl, first, k, g1, g2):
    """PS -> '.3f' quantise -> rank-sum for rows [first, first + k) of one shard (with its halo rows).

    With the HIP Context everything stays in HBM: the counts shard is uploaded once, the PS shard
    is produced, quantised and consumed in place, and only the 29 B per junction of statistics
    come back.  (Engines without the device entry points -- the CPU test double of the multi-rank
    logic -- go through the same three calls on host arrays.)
    """
    if not hasattr(engine, "ps_dev"):
        ps = engine.quantize3(engine.ps(counts_ext, rp, cl))   # the _allPS.tsv text round trip
        return engine.ranksum(ps[first: first + k], g1, g2)
    n_ext, s = counts_ext.shape
    d_counts = engine.to_device(counts_ext, np.int32)
    d_rp, d_cl = engine.to_device(rp, np.int64), engine.to_device(cl if cl.size else np.zeros(1, np.int32), np.int32)
    d_ps = engine.empty((n_ext, s), np.float32)
    engine.set_param("ps.quantize3", 1)          # the '.3f' round trip is fused into the PS store
    try:
        engine.ps_dev(d_counts, d_rp, d_cl, None, d_ps)
    finally:
        engine.set_param("ps.quantize3", 0)
    d_g1, d_g2 = engine.to_device(g1, np.int32), engine.to_device(g2, np.int32)
    out = {name: engine.empty(k, dt) for name, dt in zip(STAT_NAMES, _STAT_DTYPES)}
    engine.ranksum_dev(d_ps.offset(first * s, (k, s)), d_g1, d_g2, out)
    res = {name: out[name].to_host() for name in STAT_NAMES}
    for a in (d_counts, d_rp, d_cl, d_ps, d_g1, d_g2, *out.values()):
        a.free()
    return res


def quant_compare_sharded(engine, comm, counts_rows, row_ptr, col, g1, g2):
    """counts_rows int32 [n, s] in output row order (every rank passes the same table; only its
    own slice is touched), CSR over rows, two column groups.

    Returns dict(tested, p, corrected, med1, med2, mean1, mean2, delta) for ALL n rows,
    identical on every rank.  `engine` provides ps / quantize3 / ranksum / bh (the HIP Context).
    """
    n = row_ptr.size - 1
    plan = shard.shard_plan(row_ptr, col, comm.world)
    part = plan[comm.rank]
    lo, hi, elo, ehi = part["own_lo"], part["own_hi"], part["ext_lo"], part["ext_hi"]
    max_rows = max(p["own_hi"] - p["own_lo"] for p in plan)
    table = np.zeros((max_rows, STAT_COLS), dtype=np.float64)
    if hi > lo:
        rp, cl = shard.local_csr(row_ptr, col, part)
        r = _shard_stats(engine, np.ascontiguousarray(counts_rows[elo:ehi]), rp, cl, lo - elo, hi - lo, g1, g2)
        k = hi - lo
        for c, name in enumerate(STAT_NAMES):
            table[:k, c] = r[name]
    gathered = comm.allgather_rows(table)
    full = np.concatenate([gathered[r][: plan[r]["own_hi"] - plan[r]["own_lo"]] for r in range(comm.world)], axis=0)
    assert full.shape[0] == n
    out = dict(tested=full[:, 0].astype(np.uint8), p=full[:, 1].copy(), z=full[:, 2].copy(),
               med1=full[:, 3].astype(np.float32), med2=full[:, 4].astype(np.float32),
               mean1=full[:, 5].astype(np.float32), mean2=full[:, 6].astype(np.float32),
               delta=full[:, 7].astype(np.float32))
    keep = np.flatnonzero(out["tested"])
    corrected = np.zeros(n, dtype=np.float64)
    if keep.size:
        corrected[keep] = engine.bh(out["p"][keep])
    out["corrected"] = corrected
    out["plan"] = plan
    return out


def pair_column_ranges(pairs, world):
    """Pair columns owned by each rank for the column-wise BH: contiguous, near-equal."""
    return [(q * pairs // world, (q + 1) * pairs // world) for q in range(world)]


def pairwise_sharded(engine, comm, counts_rows, row_ptr, col, correction="pairwise"):
    """`pairwise` with junction rows sharded over ranks.

    counts_rows int32 [n, s] in row order (every rank passes the same table and touches only its
    slice), CSR over rows.  Every rank computes the exclusion sums and the s(s-1)/2 Fisher
    p-values of ITS rows.  The reference's default correction is Benjamini-Hochberg down every
    pair COLUMN over ALL junctions (pairwise_fisher.py:187-191), so the p-value matrix is
    transposed across ranks: all-to-all (rows -> columns), column BH on complete columns,
    all-to-all back.  Returns dict(p=corrected [k, pairs] for this rank's own rows, own=(lo, hi),
    plan=...).  correction: "pairwise" | "none" ("all" ranks the whole n x pairs matrix at once and
    is not sharded here -- run it on one GPU).
    """
    if correction not in ("pairwise", "none"):
        raise NotImplementedError("sharded pairwise supports --multiple_test_correction pairwise|none")
    n, s = counts_rows.shape
    pairs = s * (s - 1) // 2
    plan = shard.shard_plan(row_ptr, col, comm.world)
    part = plan[comm.rank]
    lo, hi, elo, ehi = part["own_lo"], part["own_hi"], part["ext_lo"], part["ext_hi"]
    k = hi - lo
    if k > 0:
        rp, cl = shard.local_csr(row_ptr, col, part)
        ext = np.ascontiguousarray(counts_rows[elo:ehi])
        excl = engine.ps(ext, rp, cl, want_excl=True, want_ps=False)
        p = engine.fisher_pairs(ext[lo - elo: hi - elo], excl[lo - elo: hi - elo])
    else:
        p = np.zeros((0, pairs), dtype=np.float64)
    if correction == "pairwise" and pairs > 0:
        ranges = pair_column_ranges(pairs, comm.world)
        rows_of = [q["own_hi"] - q["own_lo"] for q in plan]
        maxk = max(max(rows_of), 1)
        maxw = max(max(b - a for a, b in ranges), 1)
        blocks = []
        for a, b in ranges:                                   # my rows of rank q's columns
            blk = np.zeros((maxk, maxw), dtype=np.float64)
            blk[:k, : b - a] = p[:, a:b]
            blocks.append(blk)
        got = comm.alltoall(blocks)                           # rank r's rows of MY columns
        a, b = ranges[comm.rank]
        mine = np.concatenate([got[r][: rows_of[r], : b - a] for r in range(comm.world)], axis=0)
        assert mine.shape == (n, b - a)
        if mine.size:
            mine = engine.bh_columns(mine)
        back, at = [], 0
        for r in range(comm.world):                           # corrected values go home
            blk = np.zeros((maxk, maxw), dtype=np.float64)
            blk[: rows_of[r], : b - a] = mine[at: at + rows_of[r]]
            at += rows_of[r]
            back.append(blk)
        got = comm.alltoall(back)                             # my rows of rank q's columns, corrected
        for q, (a, b) in enumerate(ranges):
            p[:, a:b] = got[q][:k, : b - a]
    return dict(p=p, own=(lo, hi), plan=plan)
