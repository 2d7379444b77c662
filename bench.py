#!/usr/bin/env python3
"""Benchmark of the MI355X splicedice hot path (driver contract: one JSON line on rank 0).

    python bench.py --gpus 1 --steps K --warmup W              # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W  # N ranks, one per GPU

Workloads (BASELINE.json configs):
  quant   (default) : 2M junctions x 500 samples int32 counts resident in HBM -- the size north_star quotes its target
          on ("a synthetic 2M-junction x 500-sample count matrix"); BASELINE config 2 (1M x 100) rides under "also".
          One step = the whole quant device path on the rank's junction shard:
          sdice_cluster_dev (sort + overlap lists) + sdice_ps_dev (exclusion sums + PS).
          metric = PS-matrix entries/s.
  compare (config 3)          : 1M rows x (50 v 50) float32 PS table; step = sdice_ranksum_dev
          + sdice_bh_dev; metric = junction tests/s.
  pairwise (config 4 per-GPU shard) : 25k junctions x 200 samples; step = exclusion sums +
          sdice_fisher_pairs_dev + BH per pair column; metric = p-values/s.
  e2e     (config 5 per-GPU shard)  : 625k junctions x 1000 samples; step = cluster + PS + quantise
          + rank-sum (500 v 500) + BH; metric = PS entries/s.
  N > 1, pairwise / e2e: ONE dataset (N x the per-GPU size; N = 8: configs 4 / 5) cut by the library's shard plan;
          the step is the product's sharded pipeline (distributed.PairwiseShard / CompareShard): device-side packing,
          RCCL all-to-all twice + column BH (pairwise), ONE RCCL all-gather of the packed per-junction table + BH (e2e).

Multi-GPU (quant): ONE dataset of N x 1 M junctions is cut into N row ranges at chromosome boundaries
(zero halo; the ranges differ by a fraction of a per cent) and every rank clusters + quantifies its own
range: per-GPU work fixed as N grows, "scaling": "weak" -- but over one junction set and real cuts, not N
independent copies.  --strong keeps the TOTAL fixed instead (the N = 1 dataset cut into N ranges).  No
data-path collective is needed inside a step; the all-gather of the PS shards that the north star names is
timed after the loop and reported in "ps_allgather" beside the design that leaves the shards in their
owners' HBM.  torch.distributed (gloo) is the control plane only (barrier, max / sum over
ranks, the 128-byte RCCL id); the collectives are the library's own RCCL calls.

The timed region is bracketed by barrier + device sync on both sides; rank 0 prints the
max-over-ranks time.  "roofline" is for the dominant kernel, timed with HIP events on the
library's stream inside the timed region (profiling mode 2 records only that kernel).
"cpu_baseline" times the oracle's loop-for-loop restatement of the reference on a bounded
sample, on rank 0 at N=1 only; it is a reported baseline, never the thing measured above.

At N=1 the line also carries "also": the same measurement (value, roofline, cpu_baseline, verify) for
BASELINE configs 2 (quant 1M x 100), 3, 4 (one GPU's shard) and 5 (one GPU's shard), a few steps each
(--no-also skips them).  The big tables (the headline's included) are one seeded host block of 200 000 rows
repeated down the device matrix, so that generation stays bounded.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# The library itself is dlopen'ed when the first Context is created.  At N=1 torch is never
# imported (HIP runtime of /opt/rocm); at N>1 torch.distributed (gloo control plane) is imported
# first and the library then binds to the HIP runtime / librccl that torch ships -- both orders
# were run on MI355X.
from splicedice_amd.engine import Context  # noqa: E402
from splicedice_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0        # same guide: measured float4-copy ceiling (SURVEY 8(d) asks for both denominators)
VALU_WAVE_INSTS_SPEC = 6.14e11   # 1024 SIMDs x 2.4 GHz (the guide's peak engine clock) / 4 cycles per wave64 VALU instruction, f64 included.
                                 # It is a ceiling on paper: under load the chip holds a lower clock, so no kernel reaches 1.0 of it (a dependent
                                 # f64 FMA loop, tools/mb/microbench.hip, measured 0.85; that loop is NOT a ceiling and is no longer quoted)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # (a 2 ms step at the default size; short runs end before the clocks have settled)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["quant", "compare", "pairwise", "e2e"], default="quant")
    # (long names only: under torch.distributed.run a short "--n" is swallowed by the launcher's parser)
    ap.add_argument("--junctions", dest="n", type=int, default=0, help="junctions per GPU (default: the BASELINE config)")
    ap.add_argument("--samples", dest="s", type=int, default=0, help="samples (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="junctions in the CPU-baseline sample")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configurations reported under 'also'")
    ap.add_argument("--strong", action="store_true",
                    help="N>1, quant: keep the TOTAL work fixed (the N = 1 dataset cut into N row ranges) instead of the "
                         "default: one dataset of N x the per-GPU size (per-GPU work fixed)")
    ap.add_argument("--weak", action="store_true", help="(the default at N>1; kept for old command lines)")
    return ap.parse_args()


class Dist:
    """Control plane only: rendezvous, barrier, max over ranks, one small broadcast."""

    def __init__(self, world):
        self.world = world
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.pg = None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
            assert dist.get_world_size() == world, (dist.get_world_size(), world)
            self.pg = dist

    def barrier(self):
        if self.pg:
            self.pg.barrier()

    def sum(self, x):
        if not self.pg:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.SUM)
        return float(t[0])

    def all_ok(self, ok):
        """True iff `ok` holds on every rank (keeps the ranks in step when one of them failed a step)"""
        if not self.pg:
            return bool(ok)
        import torch
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MIN)
        return bool(t[0] > 0.5)

    def max(self, x):
        if not self.pg:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return float(t[0])

    def bcast_bytes(self, b, n):
        if not self.pg:
            return b
        import torch
        t = torch.frombuffer(bytearray(b if self.rank == 0 else bytes(n)), dtype=torch.uint8).clone()
        self.pg.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def close(self):
        if self.pg:
            self.pg.destroy_process_group()


def device_tiled(ctx, make_block, n, s, dtype, block_rows):
    """[n, s] device matrix = one seeded host block of `block_rows` rows repeated down the rows
    (bounded generation time for the multi-GB tables); -> (DeviceArray, host block)"""
    blk = np.ascontiguousarray(make_block(min(n, block_rows)), dtype=dtype)
    d = ctx.empty((n, s), dtype)
    for a in range(0, n, blk.shape[0]):
        b = min(n, a + blk.shape[0])
        d.offset(a * s, (b - a, s)).upload(blk[: b - a])
    return d, blk


# ------------------------------------------------------------------------------------ workloads
class QuantWorkload:
    name = "quant: cluster + PS"
    metric = "PS-matrix entries/sec (junctions x samples)"
    unit = "entries/s"
    dtype = "int32 counts -> f32 PS"
    kernel = "ps_tile_v3_kernel"

    DEFAULT_N, DEFAULT_S, DEFAULT_BLOCK = 2_000_000, 500, 200_000      # north_star's target size

    def __init__(self, ctx, rank, n, s, block_rows=0):
        if not n and not s and not block_rows:
            block_rows = self.DEFAULT_BLOCK
        self.ctx, self.n, self.s = ctx, n or self.DEFAULT_N, s or self.DEFAULT_S
        n, s = self.n, self.s
        t = time.time()
        self.junc = synth.make_junctions(n, 2 + 1000 * rank)          # this rank's chromosome group
        self.counts_host_seed = 20 + 1000 * rank
        self.d_j = [ctx.to_device(x) for x in self.junc]
        self.d_row_of, self.d_row_ptr = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
        # rows already in output order; block_rows > 0: one seeded block repeated (see device_tiled)
        self.d_counts, counts = device_tiled(ctx, lambda m: synth.make_counts(m, s, self.counts_host_seed), n, s, np.int32,
                                             block_rows or n)
        self.tiled = counts.shape[0] < n
        self.gen_s = time.time() - t
        self.sample_counts = counts[: min(n, 200_000)].copy()
        del counts
        self.d_ps = ctx.empty((n, s), np.float32)
        self.nnz = 0
        self.units = n * s
        # algorithmic bytes per launch of the dominant kernel: 4 B count + 4 B PS per entry
        self.alg_bytes = 8.0 * n * s

    def step(self):
        # the first call is synchronous (it sizes the neighbour-list buffer and reports nnz); after that
        # the whole chain -- sort, row order, lists, PS -- is enqueued without a host round trip, and
        # validation / capacity errors of the clustering surface at the next sync (sdice.h)
        d_col, nnz = self.ctx.cluster_dev(*self.d_j, self.d_row_of, self.d_row_ptr, sync=not self.nnz)
        if nnz is not None:
            self.d_col, self.nnz = d_col, nnz
        self.ctx.ps_dev(self.d_counts, self.d_row_ptr, self.d_col, None, self.d_ps)

    def describe(self):
        tag = "BASELINE config 2" if (self.n, self.s) == (1_000_000, 100) else \
            "north_star target size" if (self.n, self.s) == (2_000_000, 500) else "custom size"
        return {"workload": f"quant {self.n} junctions x {self.s} samples per GPU ({tag}), cluster+PS",
                "junctions_per_gpu": self.n, "samples": self.s, "avg_overlap_degree": round(self.nnz / self.n, 2),
                "counts": f"one seeded block of {self.sample_counts.shape[0] if self.tiled else self.n} rows"
                          + (" repeated down the matrix" if self.tiled else "")}

    def verify(self):
        """PS of the first rows against the oracle (same CSR prefix, vectorised restatement)."""
        from oracle import oracle_np as O
        big = self.sample_counts.shape[0]            # rows whose counts are kept on the host
        rp = self.d_row_ptr.offset(0, (big + 1,)).to_host()
        col = self.d_col.offset(0, (int(rp[-1]),)).to_host()
        inside = np.minimum.reduceat(np.r_[col, 0] < big, np.minimum(rp[:-1], col.size)) | (np.diff(rp) == 0)
        want, _ = O.calculate_psi_vectorised(self.sample_counts, rp, np.minimum(col, big - 1))
        got = self.d_ps.offset(0, (big, self.s)).to_host()
        rows = np.flatnonzero(inside)[:50_000]       # rows with every neighbour among the kept rows
        return bool(np.array_equal(got[rows], want[rows], equal_nan=True)), int(rows.size)

    def cpu_baseline(self, sample):
        """Loop-for-loop restatement (oracle) of getClusters + calculatePsi on `sample` junctions, 1 core."""
        from oracle import oracle_np as O
        m = min(self.n, sample or max(20_000, 100_000_000 // self.s))     # ~1e8 entries: ~11 s of single-core work
        # a junction set of its own with the same gene layout (a prefix of the shuffled 1M set
        # would be 25x sparser and have almost no overlaps)
        cr, l, r, st = synth.make_junctions(m, 7)
        counts = np.resize(self.sample_counts, (m, self.s))
        t = time.time()
        row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
        O.calculate_psi(counts, row_ptr, col)
        dt = time.time() - t
        return {"value": m * self.s / dt, "unit": self.unit, "cores": 1, "kind": "port",
                "sample": f"{m} junctions (same gene layout, avg degree ~7) x {self.s} samples: oracle get_clusters + calculate_psi "
                          f"(Python loops + numpy as the reference), {dt:.1f} s"}


class ShardedQuantWorkload(QuantWorkload):
    """N > 1: ONE junction set cut into N row ranges at chromosome boundaries (contiguous chromosome ranges
    with near-equal junction counts; rows are in (chrom, ...) order, so no overlap edge crosses a cut: zero
    halo).  Every rank clusters and quantifies its own range.  Default: the set has N x the per-GPU size
    (per-GPU work fixed: weak scaling); `strong`: the N = 1 set (total work fixed)."""
    name = "quant: cluster + PS, one dataset sharded over the ranks"

    def __init__(self, ctx, rank, world, n, s, strong=False):
        self.ctx, self.s = ctx, s or self.DEFAULT_S
        self.strong = strong
        self.n_total = (n or self.DEFAULT_N) * (1 if strong else world)
        s = self.s
        t = time.time()
        cr, l, r, st = synth.make_junctions(self.n_total, 2)          # the N = 1 dataset, identical on every rank
        self.ranges = synth.chrom_ranges(cr, world)
        c_lo, c_hi, self.row_lo, self.row_hi = self.ranges[rank]
        mine = (cr >= c_lo) & (cr < c_hi)
        self.junc = tuple(np.ascontiguousarray(x[mine]) for x in (cr, l, r, st))
        self.n = n_own = int(mine.sum())
        assert n_own == self.row_hi - self.row_lo
        # rows of the shared table, output order: row r of the table is row r mod 200 000 of one seeded block (bounded
        # generation time; every rank derives the same table)
        blk = synth.make_counts(min(self.n_total, self.DEFAULT_BLOCK), s, 20)
        counts = blk[np.arange(self.row_lo, self.row_hi) % blk.shape[0]]
        del blk
        self.gen_s = time.time() - t
        self.tiled = self.n_total > self.DEFAULT_BLOCK
        self.d_j = [ctx.to_device(x) for x in self.junc]
        self.d_row_of, self.d_row_ptr = ctx.empty(max(n_own, 1), np.int32), ctx.empty(n_own + 1, np.int64)
        self.d_counts = ctx.to_device(counts if n_own else np.zeros((1, s), np.int32))
        self.sample_counts = counts[: min(n_own, 200_000)].copy()
        del counts
        self.d_ps = ctx.empty((max(n_own, 1), s), np.float32)
        self.nnz = 0
        self.units = n_own * s
        self.alg_bytes = 8.0 * n_own * s
        self.world = world

    def describe(self):
        d = super().describe()
        d["workload"] = (f"quant {self.n_total} junctions x {self.s} samples in total "
                         f"({'the N = 1 dataset' if self.strong else 'N x the N = 1 dataset'}), "
                         f"one junction set cut at chromosome boundaries into {self.world} row ranges, cluster+PS on each")
        d["junctions_total"] = self.n_total
        d["rows_per_rank"] = [b[3] - b[2] for b in self.ranges]
        return d

    def ps_allgather(self, dist, reps=3):
        """the collective BASELINE.json's north_star names -- all-gather of the PS shards over xGMI --
        timed beside the design that leaves every shard in its owner's HBM.  Every rank goes through the
        same sequence of control-plane calls whatever fails where (a rank that raised must not leave the
        others waiting in a barrier)."""
        from splicedice_amd import distributed
        err, comm = None, None
        try:
            comm = distributed.RcclComm(self.ctx, dist.rank, self.world, dist.bcast_bytes)
        except Exception as e:
            err = f"communicator: {str(e)[:240]}"
        if not dist.all_ok(err is None):
            return {"ok": False, "error": err or "communicator failed on another rank"}
        max_rows = max(b[3] - b[2] for b in self.ranges)
        ms, ok = float("nan"), False
        try:
            out = distributed.gather_ps_dev(self.ctx, comm, self.d_ps, self.n, self.s, max_rows)
            self.ctx.sync()
            self.ctx.timer_start()
            for _ in range(reps):
                out = distributed.gather_ps_dev(self.ctx, comm, self.d_ps, self.n, self.s, max_rows)
            ms = self.ctx.timer_stop() / reps
            got = out.offset(dist.rank * max_rows * self.s, (min(self.n, 64), self.s)).to_host()
            ok = bool(np.array_equal(got, self.d_ps.offset(0, (min(self.n, 64), self.s)).to_host(), equal_nan=True))
        except Exception as e:
            err = f"all-gather: {str(e)[:240]}"
        if not dist.all_ok(err is None):
            return {"ok": False, "error": err or "all-gather failed on another rank"}
        total = self.world * max_rows * self.s * 4
        ms = dist.max(ms)
        return {"ok": dist.all_ok(ok), "ms": round(ms, 4), "bytes_gathered_per_rank": total, "rccl_ranks": self.world,
                "GB_per_s_per_rank": round(total * (self.world - 1) / self.world / (ms * 1e-3) / 1e9, 1)}


class CompareWorkload:
    name = "compare_sample_sets: rank-sum + BH"
    metric = "junction rank-sum tests/sec (50 v 50)"
    unit = "rows/s"
    dtype = "f32 PS -> f64 p"
    kernel = "ranksum_pairq_kernel"

    def __init__(self, ctx, rank, n, s, block_rows=0):
        self.ctx, self.n, self.s = ctx, n or 1_000_000, s or 100
        n, s = self.n, self.s
        t = time.time()
        self.d_ps, ps = device_tiled(ctx, lambda m: synth.make_ps_matrix(m, s, 3 + 1000 * rank), n, s, np.float32,
                                     block_rows or n)
        self.gen_s = time.time() - t
        self.g1 = np.arange(0, s // 2, dtype=np.int32)
        self.g2 = np.arange(s // 2, s, dtype=np.int32)
        self.sample_ps = ps[: min(n, 100_000)].copy()
        del ps
        self.d_g1, self.d_g2 = ctx.to_device(self.g1), ctx.to_device(self.g2)
        self.out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64),
                        med1=ctx.empty(n, np.float32), med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32),
                        mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
        self.d_q = ctx.empty(n, np.float64)
        self.units = n
        if max(self.g1.size, self.g2.size) > 64:
            self.kernel = "ranksum_count_kernel" if max(self.g1.size, self.g2.size) <= 1024 else "ranksum_block_kernel"
        self.alg_bytes = (4.0 * s + 28.0) * n          # 4*S_sel + 28 B per junction

    def step(self):
        self.ctx.ranksum_dev(self.d_ps, self.d_g1, self.d_g2, self.out)
        self.ctx.bh_masked_dev(self.out["p"], self.out["tested"], self.d_q)     # BH over the tested rows, as the product does

    def describe(self):
        return {"workload": f"compare_sample_sets {self.n} junctions, {self.g1.size} v {self.g2.size} (BASELINE config 3), "
                            f"rank-sum + BH", "junctions_per_gpu": self.n, "samples": self.s}

    def verify(self):
        from oracle import oracle_np as O
        m = min(self.n, 3000)
        want = O.compare_rows(self.sample_ps[:m], self.g1, self.g2)
        got = {k: v.to_host()[:m] for k, v in self.out.items()}
        t = want["tested"].astype(bool)
        ok = np.array_equal(got["tested"], want["tested"]) and np.array_equal(got["z"][t], want["z"][t]) \
            and np.allclose(got["p"][t], want["p"][t], rtol=1e-9, atol=0) \
            and all(np.array_equal(got[k][t], want[k][t]) for k in ("med1", "med2", "mean1", "mean2", "delta"))
        return bool(ok), m

    def cpu_baseline(self, sample):
        from oracle import oracle_np as O
        m = min(self.n, sample or 100_000, self.sample_ps.shape[0])     # ~16 s of single-core work
        t = time.time()
        r = O.compare_rows(self.sample_ps[:m], self.g1, self.g2)
        O.bh_fdr(r["p"][r["tested"].astype(bool)])
        dt = time.time() - t
        return {"value": m / dt, "unit": self.unit, "cores": 1, "kind": "port",
                "sample": f"first {m} rows: oracle compare_rows (scipy.stats.ranksums + np.median/np.mean per row, "
                          f"as compareSampleSets.py:216-232) + BH, {dt:.1f} s"}


class PairwiseWorkload:
    name = "pairwise: exclusion sums + Fisher exact"
    metric = "Fisher exact p-values/sec (all sample pairs)"
    unit = "p-values/s"
    dtype = "int64 tables -> f64 p"
    kernel = "fisher_pairs_kernel"

    def __init__(self, ctx, rank, n, s, block_rows=0):
        # config 4 is 200k junctions x 200 samples sharded over 8 GPUs -> 25k junctions per GPU
        self.ctx, self.n, self.s = ctx, n or 25_000, s or 200
        n, s = self.n, self.s
        t = time.time()
        self.junc = synth.make_junctions(n, 4 + 1000 * rank)
        counts_in = synth.make_counts(n, s, 40 + 1000 * rank)
        self.gen_s = time.time() - t
        row_of, self.row_ptr, self.col = ctx.cluster(*self.junc)
        self.counts = np.zeros_like(counts_in)
        self.counts[row_of] = counts_in
        self.d_counts = ctx.to_device(self.counts)
        self.d_row_ptr, self.d_col = ctx.to_device(self.row_ptr), ctx.to_device(self.col)
        self.d_excl = ctx.empty((n, s), np.int64)
        self.pairs = s * (s - 1) // 2
        self.d_p = ctx.empty((n, self.pairs), np.float64)
        self.units = n * self.pairs
        self.alg_bytes = 8.0 * n * self.pairs + 12.0 * n * s     # 8 B per p-value + inputs once

    def step(self):
        # the reference's default pipeline: exclusion sums, Fisher per pair, BH down every pair column
        self.ctx.ps_dev(self.d_counts, self.d_row_ptr, self.d_col, self.d_excl, None)
        self.ctx.fisher_pairs_dev(self.d_counts, self.d_excl, self.d_p)
        self.ctx.bh_columns_dev(self.d_p)

    def describe(self):
        return {"workload": f"pairwise {self.n} junctions x {self.s} samples per GPU = {self.pairs} pairs/junction "
                            f"(BASELINE config 4 is 200k junctions over 8 GPUs): exclusion sums + Fisher + "
                            f"BH per pair column (the default --multiple_test_correction)", "junctions_per_gpu": self.n,
                "samples": self.s}

    def verify(self):
        from oracle import oracle_np as O
        m = 2
        cols = 12
        self.ctx.fisher_pairs_dev(self.d_counts, self.d_excl, self.d_p)     # raw p-values again (d_p holds BH output)
        excl = self.d_excl.offset(0, (m, self.s)).to_host()[:, :cols]
        want = O.fisher_pairs(self.counts[:m, :cols], excl)
        p = self.d_p.offset(0, (m, self.pairs)).to_host()
        idx = [i * self.s - i * (i + 1) // 2 + (j - i - 1) for i in range(cols - 1) for j in range(i + 1, cols)]
        ok = np.allclose(p[:, idx], want, rtol=1e-9, atol=0)
        _, want_excl = O.calculate_psi_vectorised(self.counts, self.row_ptr, self.col)
        ok = ok and np.array_equal(self.d_excl.to_host(), want_excl)
        # the corrected table (the step's output): BH of a few whole columns against the restated definition
        raw_cols = {}
        for c in (0, self.pairs // 2, self.pairs - 1):
            one = self.ctx.empty((self.n,), np.float64)
            self.ctx.copy2d_dev(one.ptr, 8, self.d_p.ptr + c * 8, self.pairs * 8, 8, self.n)
            raw_cols[c] = (one, one.to_host())
        self.ctx.bh_columns_dev(self.d_p)
        for c, (one, raw) in raw_cols.items():
            self.ctx.copy2d_dev(one.ptr, 8, self.d_p.ptr + c * 8, self.pairs * 8, 8, self.n)
            ok = ok and bool(np.allclose(one.to_host(), O.bh_fdr(raw), rtol=1e-12, atol=0))
        return bool(ok), m * len(idx) + 3 * self.n

    def cpu_baseline(self, sample):
        from oracle import oracle_np as O
        m = sample or 20             # ~8 s of single-core work
        cols = 60
        excl = self.d_excl.offset(0, (m, self.s)).to_host()[:, :cols]
        t = time.time()
        O.fisher_pairs(self.counts[:m, :cols], excl)
        dt = time.time() - t
        nt = m * cols * (cols - 1) // 2
        return {"value": nt / dt, "unit": self.unit, "cores": 1, "kind": "port",
                "sample": f"{nt} tables ({m} junctions x {cols} samples): scipy.stats.fisher_exact per pair as "
                          f"pairwise_fisher.py:164-179, {dt:.1f} s"}


class E2EWorkload(QuantWorkload):
    """BASELINE config 5, one GPU's shard: quant + compare_sample_sets end to end, device resident.

    step = cluster -> exclusion sums + PS -> '.3f' quantise (the _allPS.tsv text round trip) ->
    rank-sum (500 v 500) + medians/means -> [N>1: RCCL all-gather of the p-values] -> BH.
    The PS shard never leaves HBM; only the per-junction p-values are exchanged.
    """
    name = "quant + compare_sample_sets end to end"
    kernel = "ranksum_count_kernel"

    def __init__(self, ctx, rank, n, s, block_rows=0):
        # config 5 is 5M junctions x 1000 samples over 8 GPUs -> 625k junctions per GPU
        super().__init__(ctx, rank, n or 625_000, s or 1000, block_rows)
        n, s = self.n, self.s
        self.g1 = np.arange(0, s // 2, dtype=np.int32)
        self.g2 = np.arange(s // 2, s, dtype=np.int32)
        self.d_g1, self.d_g2 = ctx.to_device(self.g1), ctx.to_device(self.g2)
        self.out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64),
                        med1=ctx.empty(n, np.float32), med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32),
                        mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
        self.d_q = ctx.empty(n, np.float64)
        self.collective = "none (1 GPU; N > 1: ShardedE2EWorkload)"
        big = max(self.g1.size, self.g2.size)
        self.kernel = ("ranksum_pairq_kernel" if big > 16 and self.g1.size <= 63 else "ranksum_pair_kernel") if big <= 64 else "ranksum_count_kernel" if big <= 1024 else "ranksum_block_kernel"
        self.alg_bytes = (4.0 * s + 28.0) * n           # rank-sum: 4*S_sel + 28 B per junction

    def step(self):
        self.ctx.set_param("ps.quantize3", 1)           # the '.3f' round trip rides on the PS store
        try:
            super().step()
        finally:
            self.ctx.set_param("ps.quantize3", 0)
        self.ctx.ranksum_dev(self.d_ps, self.d_g1, self.d_g2, self.out)
        self.ctx.bh_masked_dev(self.out["p"], self.out["tested"], self.d_q)     # BH over the tested rows, as the product does

    def describe(self):
        return {"workload": f"quant + compare end to end, {self.n} junctions x {self.s} samples per GPU "
                            f"({self.g1.size} v {self.g2.size}; BASELINE config 5 is 5M x 1000 over 8 GPUs): cluster + PS + "
                            f"quantise + rank-sum + BH", "junctions_per_gpu": self.n, "samples": self.s,
                "avg_overlap_degree": round(self.nnz / self.n, 2), "collective": self.collective}

    def verify(self):
        from oracle import oracle_np as O
        m = 300
        ps = self.d_ps.offset(0, (m, self.s)).to_host()
        want = O.compare_rows(ps, self.g1, self.g2)          # the device's own quantised PS rows
        got = {k: v.offset(0, (m,)).to_host() for k, v in self.out.items()}
        t = want["tested"].astype(bool)
        ok = np.array_equal(got["tested"], want["tested"]) and np.array_equal(got["z"][t], want["z"][t]) \
            and np.allclose(got["p"][t], want["p"][t], rtol=1e-9, atol=0) \
            and all(np.array_equal(got[k][t], want[k][t]) for k in ("med1", "med2", "mean1", "mean2", "delta"))
        # and the PS rows themselves: quantised oracle PS of rows whose neighbours are all kept on the host
        big = min(self.sample_counts.shape[0], 4000)
        rp = self.d_row_ptr.offset(0, (big + 1,)).to_host()
        col = self.d_col.offset(0, (int(rp[-1]),)).to_host()
        inside = np.minimum.reduceat(np.r_[col, 0] < big, np.minimum(rp[:-1], col.size)) | (np.diff(rp) == 0)
        rows = np.flatnonzero(inside[:m])
        want_ps, _ = O.calculate_psi_vectorised(self.sample_counts[:big], rp, np.minimum(col, big - 1))
        ok = ok and np.array_equal(ps[rows], O.quantize3_fast(want_ps[:m])[rows], equal_nan=True)
        return bool(ok), m

    def cpu_baseline(self, sample):
        from oracle import oracle_np as O
        m = min(self.n, sample or 30_000)
        cr, l, r, st = synth.make_junctions(m, 7)
        counts = np.resize(self.sample_counts, (m, self.s))
        t = time.time()
        row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
        ps, _ = O.calculate_psi(counts, row_ptr, col)
        ps = O.quantize3_fast(ps)
        res = O.compare_rows(ps, self.g1, self.g2)
        O.bh_fdr(res["p"][res["tested"].astype(bool)])
        dt = time.time() - t
        return {"value": m * self.s / dt, "unit": self.unit, "cores": 1, "kind": "port",
                "sample": f"{m} junctions x {self.s} samples: oracle get_clusters + calculate_psi + '.3f' round trip + "
                          f"compare_rows (scipy ranksums per row) + BH, {dt:.1f} s"}


class _RefusedComm:
    """stands in for the RCCL communicator in the one-GPU rehearsal of the N>1 control flow (RCCL refuses two ranks on
    one device): blocks stay where they are, the line reports the collective as refused"""
    device = True

    def __init__(self, rank, world, why):
        self.rank, self.world, self.why = rank, world, why

    def allgather(self, x):
        out = x.ctx.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), x.dtype).zero()
        return out

    def alltoall(self, x):
        return x

    def allsum(self, v):
        return int(v)


def _device_comm(ctx, dist, world):
    """-> (communicator, note): the library's RCCL communicator; in the one-GPU rehearsal a stand-in"""
    from splicedice_amd import distributed
    err = None
    try:
        comm = distributed.RcclComm(ctx, dist.rank, world, dist.bcast_bytes)
    except Exception as e:                                   # noqa: BLE001
        comm, err = None, str(e)[:200]
    if not dist.all_ok(err is None):
        if os.environ.get("SDICE_BENCH_DEVICE") is None:
            raise RuntimeError(f"RCCL communicator: {err or 'failed on another rank'}")   # a real multi-GPU run must not lose its exchange
        return _RefusedComm(dist.rank, world, err), f"refused (one-GPU rehearsal): {err or 'failed on another rank'}"
    return comm, None


def _tiled_rows(block, lo, hi):
    """rows [lo, hi) of the table whose row r is row r mod len(block) of one seeded block (every rank derives the same table)"""
    return np.ascontiguousarray(block[np.arange(lo, hi) % block.shape[0]])


def _sharded_setup(world, rank, n_total, seed):
    """ONE junction set in output row order, cut from the COORDINATES (shard.shard_plan_junctions: no rank clusters the
    whole set) -> (plan, this rank's part, the coordinates of its rows [ext_lo, ext_hi))"""
    from splicedice_amd import shard
    junc = synth.make_junctions(n_total, seed)             # identical on every rank
    o = shard.junction_order(*junc)
    junc = tuple(np.ascontiguousarray(x[o]) for x in junc)
    plan = shard.shard_plan_junctions(*junc, world)
    part = plan[rank]
    return plan, part, tuple(np.ascontiguousarray(x[part["ext_lo"]:part["ext_hi"]]) for x in junc)


class ShardedPairwiseWorkload:
    """N > 1, `pairwise`: ONE junction set of N x 25 000 junctions x 200 samples (N = 8: BASELINE config 4) cut by the
    library's shard plan (sdice_shard_plan_junctions: clean cuts, from the coordinates); every rank holds ITS count rows
    only and clusters ITS rows only.  step = the product's sharded pipeline (distributed.PairwiseShard.step): clustering
    of the rank's range + exclusion sums + Fisher on the rank's rows, device-side packing, RCCL all-to-all (rows -> pair
    columns), BH down complete columns, all-to-all back, unpack."""
    name = "pairwise, one dataset sharded over the ranks"
    metric = PairwiseWorkload.metric
    unit = PairwiseWorkload.unit
    dtype = PairwiseWorkload.dtype
    kernel = "fisher_pairs_kernel"

    def __init__(self, ctx, dist, world, n, s):
        from splicedice_amd import distributed
        self.ctx, self.world, self.s = ctx, world, s or 200
        self.n_total = (n or 25_000) * world
        s = self.s
        t = time.time()
        self.plan, part, self.jext = _sharded_setup(world, dist.rank, self.n_total, 4)
        blk = synth.make_counts(min(self.n_total, 50_000), s, 40)
        ext = _tiled_rows(blk, part["ext_lo"], part["ext_hi"])
        self.gen_s = time.time() - t
        self.comm, self.comm_note = _device_comm(ctx, dist, world)
        self.shard = distributed.PairwiseShard(ctx, self.comm, self.n_total, s, self.plan, "pairwise", "fisher")
        self.shard.load(ext, junctions=self.jext)
        self.n = part["own_hi"] - part["own_lo"]
        self.pairs = s * (s - 1) // 2
        self.units = self.n * self.pairs
        self.alg_bytes = 8.0 * self.n * self.pairs + 12.0 * self.n * s
        self.ext, self.part = ext, part

    def step(self):
        self.shard.step()

    def describe(self):
        return {"workload": f"pairwise {self.n_total} junctions x {self.s} samples in total over {self.world} ranks "
                            f"(BASELINE config 4 is 200k junctions over 8 GPUs), one junction set cut by sdice_shard_plan_junctions: "
                            f"cluster (the rank's own range) + exclusion sums + Fisher + pack + all-to-all + BH per pair column + "
                            f"all-to-all back",
                "junctions_total": self.n_total, "rows_per_rank": [q["own_hi"] - q["own_lo"] for q in self.plan],
                "samples": self.s, "collective": self.comm_note or "RCCL all-to-all (grouped send/recv), twice per step",
                "collectives": self.shard.timed_collectives() if self.comm_note is None else None}

    def verify(self):
        """raw Fisher p-values of a few of the rank's rows against the oracle (the corrected matrix needs every rank's rows)"""
        from oracle import oracle_np as O
        from splicedice_amd import distributed
        sh = distributed.PairwiseShard(self.ctx, distributed.SingleComm(), self.n_total, self.s, self.plan[self.comm.rank: self.comm.rank + 1],
                                       "none", "fisher")
        ok = True
        try:
            sh.load(self.ext, junctions=self.jext)
            sh.step()
            p = sh.result()
            rp, cl, _ = sh.csr_host()
            a0 = self.part["own_lo"] - self.part["ext_lo"]
            m, cols = min(2, self.n), 12
            _, excl = O.calculate_psi_vectorised(self.ext, rp, cl)
            want = O.fisher_pairs(self.ext[a0: a0 + m, :cols], excl[a0: a0 + m, :cols])
            idx = [i * self.s - i * (i + 1) // 2 + (j - i - 1) for i in range(cols - 1) for j in range(i + 1, cols)]
            ok = bool(np.allclose(p[:m][:, idx], want, rtol=1e-9, atol=0))
        finally:
            sh.free()
        return ok, 2 * 66

    def cpu_baseline(self, sample):
        return None


class ShardedE2EWorkload:
    """N > 1, quant + compare end to end: ONE junction set of N x 625 000 junctions x 1000 samples (N = 8: BASELINE
    config 5) cut by the library's shard plan (from the coordinates: no rank clusters the whole set).  step = the
    product's sharded pipeline (distributed.CompareShard.step): clustering of the rank's OWN rows [ext_lo, ext_hi)
    (asynchronous, per-rank work that shrinks with N), PS with the '.3f' round trip on them, rank-sum into ONE packed
    per-junction block, ONE RCCL all-gather of that block (37 B per junction), BH over the gathered p-values."""
    name = "quant + compare_sample_sets end to end, one dataset sharded over the ranks"
    metric = QuantWorkload.metric
    unit = QuantWorkload.unit
    dtype = QuantWorkload.dtype
    kernel = "ranksum_count_kernel"

    def __init__(self, ctx, dist, world, n, s):
        from splicedice_amd import distributed
        self.ctx, self.world, self.s = ctx, world, s or 1000
        self.n_total = (n or 625_000) * world
        s, nt = self.s, self.n_total
        t = time.time()
        self.plan, part, jext = _sharded_setup(world, dist.rank, nt, 5)
        blk = synth.make_counts(min(nt, 125_000), s, 20)
        ext = _tiled_rows(blk, part["ext_lo"], part["ext_hi"])
        del blk
        self.gen_s = time.time() - t
        self.g1, self.g2 = np.arange(0, s // 2, dtype=np.int32), np.arange(s // 2, s, dtype=np.int32)
        self.comm, self.comm_note = _device_comm(ctx, dist, world)
        self.shard = distributed.CompareShard(ctx, self.comm, nt, s, self.plan, self.g1, self.g2)
        self.shard.load(ext, junctions=jext)
        self.n = part["own_hi"] - part["own_lo"]
        self.units = self.n * s
        self.alg_bytes = (4.0 * s + 28.0) * self.n
        big = max(self.g1.size, self.g2.size)
        self.kernel = ("ranksum_pairq_kernel" if big > 16 and self.g1.size <= 63 else "ranksum_pair_kernel") if big <= 64 else "ranksum_count_kernel" if big <= 1024 else "ranksum_block_kernel"
        self.part = part
        self.nnz_own = None
        del ext

    def step(self):
        self.shard.step()

    def describe(self):
        _, block = self.shard.off, self.shard.block
        if self.nnz_own is None:
            self.nnz_own = self.shard.csr_host()[2]
        rows = self.part["ext_hi"] - self.part["ext_lo"]
        return {"workload": f"quant + compare end to end, {self.n_total} junctions x {self.s} samples in total over {self.world} ranks "
                            f"(BASELINE config 5 is 5M x 1000 over 8 GPUs), one junction set cut by sdice_shard_plan_junctions: cluster "
                            f"(the rank's own range) + PS + quantise + rank-sum + ONE all-gather of the per-junction table + BH",
                "junctions_total": self.n_total, "rows_per_rank": [q["own_hi"] - q["own_lo"] for q in self.plan],
                "samples": self.s, "avg_overlap_degree": round(self.nnz_own / max(rows, 1), 2),
                "collective": self.comm_note or f"ONE RCCL all-gather of {block} B per rank (the packed per-junction table) per step"}

    def verify(self):
        from oracle import oracle_np as O
        m = min(300, self.n)
        first = self.part["own_lo"] - self.part["ext_lo"]
        ps = self.shard.d_ps.offset(first * self.s, (m, self.s)).to_host()
        want = O.compare_rows(ps, self.g1, self.g2)          # the device's own quantised PS rows
        got = {name: v.offset(0, (m,)).to_host() for name, v in self.shard.views.items()}
        t = want["tested"].astype(bool)
        ok = np.array_equal(got["tested"], want["tested"]) and np.array_equal(got["z"][t], want["z"][t]) \
            and np.allclose(got["p"][t], want["p"][t], rtol=1e-9, atol=0) \
            and all(np.array_equal(got[k][t], want[k][t]) for k in ("med1", "med2", "mean1", "mean2", "delta"))
        return bool(ok), m

    def cpu_baseline(self, sample):
        return None


WORKLOADS = {"quant": QuantWorkload, "compare": CompareWorkload, "pairwise": PairwiseWorkload, "e2e": E2EWorkload}


KERNEL_SOURCES = {"quant": ["ps.hip"], "compare": ["ranksum.hip"], "pairwise": ["fisher.hip"], "e2e": ["ranksum.hip"]}


def kernel_source_sha16(workload):
    """sha256 (first 16 hex digits) of the source file(s) of the workload's dominant kernel: a PMC pass in
    profiles/pmc_traffic.json is only quoted while the kernel it measured is the kernel that runs"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[workload]:
        with open(os.path.join(REPO, "splicedice_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_from_profiles(workload, n, s):
    """(HBM bytes per launch of the dominant kernel or None, provenance) from the committed rocprofv3 PMC passes:
    the counters are collected in runs of their own (profiles/), not while this line is measured.  A pass taken on
    an older version of the kernel's source is NOT quoted (traffic: null, the note says why)."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        sha = kernel_source_sha16(workload)
        with open(path) as fh:
            for rec in json.load(fh):
                if rec["workload"] == workload and rec["n"] == n and rec["s"] == s:
                    if rec.get("src_sha16") != sha:
                        return None, (f"stale profile: {rec.get('source', path)} was taken on source {rec.get('src_sha16', '(unstamped)')}, "
                                      f"the kernel source is now {sha}")
                    return rec["hbm_bytes_per_launch"], "committed PMC pass: " + rec.get("source", path)
    except (OSError, ValueError, KeyError):
        pass
    return None


MP_SAMPLE = {"quant": (400_000, 100), "compare": (15_000, 100), "pairwise": (8, 60), "e2e": (16_000, 1000)}


def cpu_baseline_all_cores(workload):
    """SURVEY 8(d)'s second CPU figure: the same restatement junction-sharded over the host cores
    (multiprocessing), run as a child process BEFORE this process touches the GPU."""
    import subprocess
    m, s = MP_SAMPLE[workload]
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)       # a one-GPU box's CPU share
    cmd = [sys.executable, os.path.join(REPO, "oracle", "cpu_baseline_mp.py"), "--workload", workload,
           "--units-per-core", str(m), "--samples", str(s), "--cores", str(cores)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, check=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": d["value"], "cores": d["cores"],
                "sample": f"{d['cores']} workers x {m} junctions x {s} samples each (own seeded shard), {d['seconds']} s"}
    except Exception as e:      # a reported extra, never allowed to take the GPU measurement down
        return {"value": None, "cores": cores, "error": str(e)[:200]}


def reference_over_port(workload):
    """Speed of the REAL reference relative to the port timed here, measured in the build container
    where the reference is importable (tests/golden/time_reference.py -> profiles/reference_in_container.json):
    the port is a little slower than the code it restates, so GPU/port ratios overstate GPU/reference by this factor."""
    try:
        with open(os.path.join(REPO, "profiles", "reference_in_container.json")) as fh:
            d = json.load(fh)
        if workload in ("quant", "e2e"):
            q = d["quant_100k_x_100"]
            return round(q["reference"]["entries_per_s"] / q["oracle_port"]["entries_per_s"], 2)
        if workload == "compare":
            c = d["compare_20k_rows_50v50"]
            return round(c["reference_loop_rows_per_s"] / c["oracle_port_rows_per_s"], 2)
        f = [v for k, v in d.items() if k.startswith("pairwise_fisher")][0]
        return round(f["reference_loop_p_per_s"] / f["oracle_port_p_per_s"], 2)
    except (OSError, ValueError, KeyError, IndexError):
        return None


def measure(ctx, wl, dist, steps, warmup, gpus, verify=True):
    """warmup + timed steps of one workload -> (elapsed s [max over ranks], roofline dict, verify dict)"""
    for _ in range(warmup):
        wl.step()
    ctx.sync()
    ctx.prof_enable(2)
    ctx.prof_reset()
    dist.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    ctx.sync()                                   # (also resolves the deferred status of asynchronous clusterings)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = ctx.prof_query(wl.kernel)
    ctx.prof_enable(0)
    elapsed = dist.max(elapsed)
    avg_ms = kernel_ms / launches if launches else float("nan")
    achieved = wl.alg_bytes / (avg_ms * 1e-3) / 1e9 if launches else float("nan")
    traffic = traffic_from_profiles(wl.key, wl.n, wl.s)
    roofline = {"bound": "hbm", "kernel": wl.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBS,
                "avg_kernel_ms": avg_ms, "launches": launches, "algorithmic_bytes_per_launch": wl.alg_bytes,
                "traffic": traffic[0] if traffic else None,
                "traffic_source": traffic[1] if traffic else "no PMC pass committed for this shape"}
    if wl.key == "pairwise" and launches:
        # the Fisher kernel is f64-VALU bound: achieved VALU issue against the f64 issue rate of the chip
        # (tools/mb/microbench.hip: 33.6e12 f64 FMA lane-operations/s = 5.25e11 wave64 instructions/s), instruction
        # counts from the committed SQ counter pass of this shape.  The walk step is branch-free (every lane steps in
        # every trip), so the hardware's active-lane count says nothing; the kernel counts its own lane-steps in one
        # extra launch outside the timed region (fisher.count_steps): issued, and those that advanced a live walk
        # inside its support.
        vf = {"peak": VALU_WAVE_INSTS_SPEC, "unit": "wave64 VALU instructions/s (all VALU, f64 and not)",
              "peak_assumes": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"}
        try:
            if not all(hasattr(wl, a) for a in ("d_counts", "d_excl", "d_p")):
                raise AttributeError("lane-step counts are taken on the unsharded workload (N = 1)")
            ctx.set_param("fisher.count_steps", 1)
            ctx.fisher_pairs_dev(wl.d_counts, wl.d_excl, wl.d_p)
            useful, issued = ctx.fisher_step_stats()
            ctx.set_param("fisher.count_steps", 0)
            vf.update({"lane_steps_issued": issued, "lane_steps_useful": useful, "useful_lane_frac": useful / max(issued, 1),
                       "useful_steps_per_pair": useful / (wl.n * (wl.s * (wl.s - 1) // 2)),
                       "valu_per_step": 10, "step_valu_rate": issued / 64 * 10 / (avg_ms * 1e-3),
                       "step_valu_frac_of_peak": issued / 64 * 10 / (avg_ms * 1e-3) / VALU_WAVE_INSTS_SPEC,
                       # ONE number: the share of the chip's peak VALU issue that advances a live walk inside its support
                       "useful_issue_frac": useful / 64 * 10 / (avg_ms * 1e-3) / VALU_WAVE_INSTS_SPEC})
        except Exception as e:                                  # noqa: BLE001 (a measurement aid must not fail the line)
            ctx.set_param("fisher.count_steps", 0)
            vf["lane_steps_note"] = str(e)
        try:
            with open(os.path.join(REPO, "profiles", "pairwise_valu.json")) as fh:
                pv = json.load(fh)
            if pv["n"] == wl.n and pv["s"] == wl.s:
                rate = pv["SQ_INSTS_VALU_per_launch"] / (avg_ms * 1e-3)
                vf.update({"achieved": rate, "frac": rate / VALU_WAVE_INSTS_SPEC,
                           "stale": pv.get("src_sha16") != kernel_source_sha16("pairwise"), "source": pv["source"]})
        except (OSError, ValueError, KeyError):
            pass
        roofline["valu_f64"] = vf
    v = None
    if verify:
        ok, checked = wl.verify()
        v = {"ok": ok, "checked": checked}
    if wl.key in ("quant", "e2e") and launches and hasattr(wl, "d_counts") and hasattr(wl, "d_ps") and hasattr(wl, "sample_counts"):
        # host <-> device rates of this workload's tables (pageable numpy buffers, as the host entry points
        # of the C ABI receive them), measured apart from the timed region and never part of `value`
        try:
            blk = wl.sample_counts
            view_in = wl.d_counts.offset(0, blk.shape)
            view_in.upload(blk)
            t0 = time.perf_counter(); view_in.upload(blk); h2d = time.perf_counter() - t0
            view_out = wl.d_ps.offset(0, blk.shape)
            view_out.to_host()
            t0 = time.perf_counter(); view_out.to_host(); d2h = time.perf_counter() - t0
            roofline["pcie"] = {"bytes": int(blk.nbytes), "h2d_GBps": blk.nbytes / h2d / 1e9, "d2h_GBps": blk.nbytes / d2h / 1e9,
                                "note": "count rows in / PS rows out through sdice_h2d / sdice_d2h, pageable host memory; "
                                        "reported beside the device-resident figures, never inside them"}
        except Exception as e:                       # noqa: BLE001
            roofline["pcie"] = {"error": str(e)[:120]}
    if wl.key == "quant" and launches and hasattr(wl, "d_counts") and hasattr(wl, "d_ps"):
        # what a device-to-device copy of the same byte volume (count table -> PS table) takes on THIS box, measured after
        # the timed region and the verification (it overwrites the PS table): the bandwidth ceiling the PS kernel can be held against besides the spec figure
        try:
            nbytes = int(wl.alg_bytes // 2)
            for _ in range(2):
                ctx.copy2d_dev(wl.d_ps.ptr, nbytes, wl.d_counts.ptr, nbytes, nbytes, 1)
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                ctx.copy2d_dev(wl.d_ps.ptr, nbytes, wl.d_counts.ptr, nbytes, nbytes, 1)
            copy_ms = ctx.timer_stop() / 5
            roofline["copy_same_bytes_ms"] = copy_ms
            roofline["frac_of_copy_same_bytes"] = copy_ms / avg_ms
        except Exception as e:                       # the figure is a reference point, never a reason to fail the line
            roofline["copy_same_bytes_ms"] = None
            roofline["copy_error"] = str(e)[:120]
    return elapsed, roofline, v


# what "also" reports at N=1: (key, workload class, n, s, rows of the repeated host block, steps, CPU sample)
ALSO = [
    ("quant_c2", "quant", 1_000_000, 100, 1_000_000, 200, 100_000),        # BASELINE config 2 (generated whole)
    ("compare_c3", "compare", 1_000_000, 100, 200_000, 30, 30_000),        # BASELINE config 3
    ("pairwise_c4_shard", "pairwise", 25_000, 200, 0, 4, 10),              # config 4, one GPU's shard of 8
    ("e2e_c5_shard", "e2e", 625_000, 1000, 125_000, 10, 12_000),            # config 5, one GPU's shard of 8
    ("pairwise_c4_full", "pairwise", 200_000, 200, 0, 2, 0),                # config 4 at its FULL size on one GPU (no CPU leg: the shard's stands)
    ("e2e_c5_full", "e2e", 5_000_000, 1000, 125_000, 3, 0),                 # config 5 at its FULL size on one GPU
]


def main():
    args = parse_args()
    all_cores = None
    if args.gpus == 1 and not args.no_cpu_baseline:
        all_cores = cpu_baseline_all_cores(args.workload)
    # stdout carries exactly ONE line (the JSON): gloo / RCCL print banners on fd 1, so fd 1 is
    # pointed at stderr for the whole run and the JSON goes to a saved copy of the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    dist = Dist(args.gpus)
    # one rank per GPU; SDICE_BENCH_DEVICE pins every rank to one device (rehearsing the N>1 control
    # flow on a one-GPU box -- RCCL then refuses the duplicate GPU and "allgather" reports it)
    forced = os.environ.get("SDICE_BENCH_DEVICE")
    ctx = Context(int(forced) if forced is not None else (dist.local_rank if args.gpus > 1 else 0))
    sharded = args.gpus > 1 and args.workload == "quant"
    strong = sharded and args.strong
    product_sharded = args.gpus > 1 and args.workload in ("pairwise", "e2e")
    if sharded:
        wl = ShardedQuantWorkload(ctx, dist.rank, args.gpus, args.n, args.s, strong=strong)
    elif product_sharded:
        # ONE dataset cut by the library's shard plan, the product's sharded pipeline inside every step
        wl = (ShardedPairwiseWorkload if args.workload == "pairwise" else ShardedE2EWorkload)(ctx, dist, args.gpus, args.n, args.s)
    else:
        wl = WORKLOADS[args.workload](ctx, dist.rank, args.n, args.s)
    wl.key = args.workload

    elapsed, roofline, verify = measure(ctx, wl, dist, args.steps, args.warmup, args.gpus, not args.no_verify)
    failed = bool(verify and not verify["ok"])

    ps_allgather = None
    if sharded:
        ps_allgather = wl.ps_allgather(dist)      # reported, never hidden: the timed steps above contain no collective
        dist.barrier()

    allgather = None
    if args.gpus > 1 and not sharded and not product_sharded:
        # data-plane collective: RCCL all-gather of a per-junction result table (8 B per junction); every rank
        # goes through the same control-plane calls whatever fails where
        err = None
        try:
            uid = ctx.comm_unique_id() if dist.rank == 0 else None
        except Exception as e:
            uid, err = bytes(128), f"unique id: {str(e)[:200]}"
        uid = dist.bcast_bytes(uid, 128)
        try:
            if err is None:
                ctx.comm_init(uid, dist.rank, args.gpus)
        except Exception as e:
            err = f"comm_init: {str(e)[:240]}"
        if not dist.all_ok(err is None):
            allgather = {"ok": False, "error": err or "communicator failed on another rank"}
        else:
            ms, ok = float("nan"), False
            try:
                send = ctx.to_device(np.full(wl.n, float(dist.rank)))
                recv = ctx.empty(wl.n * args.gpus, np.float64)
                ctx.allgather_dev(send, recv)
                ctx.sync()
                ctx.timer_start()
                for _ in range(5):
                    ctx.allgather_dev(send, recv)
                ms = ctx.timer_stop() / 5
                got = recv.to_host().reshape(args.gpus, wl.n)[:, 0]
                ok = bool(np.array_equal(got, np.arange(args.gpus)))
            except Exception as e:
                err = f"all-gather: {str(e)[:240]}"
            if not dist.all_ok(err is None):
                allgather = {"ok": False, "error": err or "all-gather failed on another rank"}
            else:
                allgather = {"ok": dist.all_ok(ok), "bytes_per_rank": wl.n * 8, "ms": round(dist.max(ms), 4),
                             "rccl_ranks": args.gpus}
        dist.barrier()

    cpu = None
    if dist.rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = wl.cpu_baseline(args.cpu_sample)
        cpu["all_cores"] = all_cores
        cpu["real_reference_over_port"] = reference_over_port(args.workload)

    total_units = dist.sum(wl.units) * args.steps          # units all ranks processed
    line = None
    if dist.rank == 0:
        info = ctx.device_info()
        line = {
            "metric": wl.metric, "value": total_units / elapsed, "unit": wl.unit, "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": wl.dtype,
            "data": "synthetic (seeded numpy PCG64; SURVEY.md 8(d))",
            "config": dict(wl.describe(), parallelism=f"junction shards x{args.gpus}, one rank per GPU"),
            "roofline": roofline, "cpu_baseline": cpu, "verify": verify, "allgather": allgather,
            "ps_allgather": ps_allgather,
            "rccl_ranks": args.gpus if any(c and c.get("ok") for c in (ps_allgather, allgather)) or
            (product_sharded and getattr(wl, "comm_note", "x") is None) else (0 if args.gpus > 1 else None),
            "device": info["name"].strip(), "gen_seconds": round(wl.gen_s, 1),
        }

    # ---- the other configurations, same measurement, a few steps each (1 GPU only: the driver's N=1 line)
    if args.gpus == 1 and not args.no_also and args.workload == "quant" and not args.n and not args.s:
        del wl
        also = {}
        for key, kind, n, s, block, steps, cpu_sample in ALSO:
            t_start = time.time()
            try:
                w = WORKLOADS[kind](ctx, 0, n, s, block)
                w.key = kind
                el, rl, vf = measure(ctx, w, dist, steps, 5, 1, not args.no_verify)      # (enough steps for the clocks to settle)
                rec = {"metric": w.metric, "value": w.units * steps / el, "unit": w.unit, "steps": steps, "warmup": 5,
                       "ms_per_step": el / steps * 1e3, "dtype": w.dtype, "config": w.describe(), "roofline": rl,
                       "verify": vf, "gen_seconds": round(w.gen_s, 1)}
                if not args.no_cpu_baseline and cpu_sample:
                    rec["cpu_baseline"] = w.cpu_baseline(cpu_sample)
                    rec["cpu_baseline"]["real_reference_over_port"] = reference_over_port(kind)
                failed = failed or bool(vf and not vf["ok"])
                del w
                import gc
                gc.collect()
                ctx.trim()                       # the scratch of a 32 GB column BH does not stay with the next record
            except Exception as e:      # an extra record must not take the headline down; it is reported as failed
                rec = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                failed = True
            rec["wall_seconds"] = round(time.time() - t_start, 1)
            also[key] = rec
        line["also"] = also

    if dist.rank == 0:
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    dist.close()
    ctx.close()
    if failed:
        sys.exit(3)              # a wrong result must not look like a successful benchmark


if __name__ == "__main__":
    main()
