#!/usr/bin/env python3
"""Per-kernel HIP-event times of sdice_bh_columns_dev in one process: prof_bh.py n cols [param=value,...] ...
cols = "fisher": the table of `bench.py --workload pairwise` (n junctions x 200 samples -> 19 900 columns of real Fisher
p-values: few distinct values per column, most of them 1) instead of random values with 20 % ones."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd.engine import Context
n = int(sys.argv[1])
cfgs = sys.argv[3:] or [""]
ctx = Context(0)
if sys.argv[2] == "fisher":
    from splicedice_amd import synth
    s = 200
    cols = s * (s - 1) // 2
    row_of, row_ptr, col = ctx.cluster(*synth.make_junctions(n, 4))
    counts_in = synth.make_counts(n, s, 40)
    counts = np.zeros_like(counts_in)
    counts[row_of] = counts_in
    d_counts, d_rp, d_col = ctx.to_device(counts), ctx.to_device(row_ptr), ctx.to_device(col)
    d_excl, d_src = ctx.empty((n, s), np.int64), ctx.empty((n, cols), np.float64)
    ctx.ps_dev(d_counts, d_rp, d_col, d_excl, None)
    ctx.fisher_pairs_dev(d_counts, d_excl, d_src)
    ctx.sync()
    if os.environ.get("BH_STATS"):
        h = d_src.to_host()
        for c in (0, 7777, cols - 1):
            v, cnt = np.unique(h[:, c], return_counts=True)
            o = np.argsort(-cnt)[:8]
            print(f"column {c}: {len(v)} distinct values of {n}; most frequent:", ", ".join(f"{v[i]:.4g} x{cnt[i]}" for i in o),
                  "; values that occur once:", int((cnt == 1).sum()), flush=True)
        del h
else:
    cols = int(sys.argv[2])
    rng = np.random.default_rng(1)
    d_src = ctx.empty((n, cols), np.float64)
    step = min(n, 2000)
    for a in range(0, n, step):                     # (fresh values per block: a repeated block would fill long columns with ties)
        b = min(n, a + step)
        blk = rng.random((b - a, cols)) ** 2
        blk[rng.random(blk.shape) < 0.2] = 1.0
        d_src.offset(a * cols, (b - a, cols)).upload(blk)
d = ctx.empty((n, cols), np.float64)
for c in cfgs:
    kv = [x.split("=") for x in c.split(",") if x]
    for k, v in kv:
        ctx.set_param(k, int(v))
    for it in range(2):
        ctx.copy2d_dev(d.ptr, cols * 8, d_src.ptr, cols * 8, cols * 8, n)
        ctx.bh_columns_dev(d)
    ctx.sync()
    ctx.prof_enable(1)
    ctx.prof_reset()
    reps = 3
    for it in range(reps):
        ctx.copy2d_dev(d.ptr, cols * 8, d_src.ptr, cols * 8, cols * 8, n)
        ctx.bh_columns_dev(d)
    ctx.sync()
    rep = ctx.prof_report()
    ctx.prof_enable(0)
    for k, v in kv:
        ctx.set_param(k, {"bh.reg_cap": 2048, "bh.mean": 0, "bh.wg": 256, "bh.big_wg": 512, "bh.spb": 8, "bh.fused_count": 1, "bh.rows_per_block": 2048, "bh.finish_cols": 16}.get(k, 0))
    tot = sum(ms for _, ms in rep.values()) / reps
    print(f"[{c}] total {tot:.3f} ms: " + ", ".join(f"{name} {ms / reps:.3f}" for name, (cnt, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])), flush=True)
