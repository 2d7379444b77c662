#!/usr/bin/env python3
"""One warm-up and `reps` calls of sdice_bh_columns_dev on n x cols (for rocprofv3 passes): run_bh_once.py n cols reps [param=value,...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd.engine import Context
n, cols, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = Context(0)
for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 and sys.argv[4] else []):
    k, v = kv.split("=")
    ctx.set_param(k, int(v))
rng = np.random.default_rng(1)
blk = rng.random((min(n, 2000), cols)) ** 2
blk[rng.random(blk.shape) < 0.2] = 1.0
d_src = ctx.empty((n, cols), np.float64)
for a in range(0, n, blk.shape[0]):
    b = min(n, a + blk.shape[0])
    d_src.offset(a * cols, (b - a, cols)).upload(np.roll(blk[: b - a], a, axis=1))
d = ctx.empty((n, cols), np.float64)
for it in range(reps + 1):
    ctx.copy2d_dev(d.ptr, cols * 8, d_src.ptr, cols * 8, cols * 8, n)
    ctx.bh_columns_dev(d)
ctx.sync()
print("done")
