#!/usr/bin/env python3
"""Same-process timing of sdice_bh_columns_dev: ab_bh.py n cols [param=value,...] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd.engine import Context
n, cols = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:] or [""]
ctx = Context(0)
rng = np.random.default_rng(1)
blk = rng.random((min(n, 2000), cols)) ** 2
blk[rng.random(blk.shape) < 0.2] = 1.0
d_src = ctx.empty((n, cols), np.float64)
for a in range(0, n, blk.shape[0]):
    b = min(n, a + blk.shape[0])
    d_src.offset(a * cols, (b - a, cols)).upload(np.roll(blk[: b - a], a, axis=1))
d = ctx.empty((n, cols), np.float64)
for rep in range(3):
    for c in cfgs:
        kv = [x.split("=") for x in c.split(",") if x]
        for k, v in kv:
            ctx.set_param(k, int(v))
        ms = []
        for it in range(4):
            ctx.copy2d_dev(d.ptr, cols * 8, d_src.ptr, cols * 8, cols * 8, n)
            ctx.sync(); ctx.timer_start(); ctx.bh_columns_dev(d); ms.append(ctx.timer_stop())
        for k, v in kv:
            ctx.set_param(k, {"bh.reg_cap": 2048, "bh.mean": 0, "bh.wg": 256, "bh.spb": 8, "bh.fused_count": 1, "bh.rows_per_block": 2048, "bh.finish_cols": 16}.get(k, 0))
        print(f"rep {rep} [{c}] {min(ms[1:]):.3f} ms  ({n * cols / min(ms[1:]) / 1e6:.2f} G values/s)", flush=True)
