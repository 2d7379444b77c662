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
