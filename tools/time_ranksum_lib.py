#!/usr/bin/env python3
"""Time the rank-sum call for 1 M rows of 50 v 50 with the library given on the command line (A/B of two builds inside
ONE gpurun call, i.e. on one GPU: `python tools/time_ranksum_lib.py build/base_lib/libsplicedice_hip.so`)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import _ffi, synth
if len(sys.argv) > 1:
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
from splicedice_amd.engine import Context
n, s = 1_000_000, 100
ctx = Context(0)
ps = synth.make_ps_matrix(200_000, s, 3)
d_ps = ctx.empty((n, s), np.float32)
for a in range(0, n, 200_000):
    d_ps.offset(a * s, (200_000, s)).upload(ps)
g1, g2 = ctx.to_device(np.arange(0, 50, dtype=np.int32)), ctx.to_device(np.arange(50, 100, dtype=np.int32))
out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64), med1=ctx.empty(n, np.float32),
           med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32), mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
for _ in range(30):
    ctx.ranksum_dev(d_ps, g1, g2, out)
ctx.sync()
res = []
for rep in range(3):
    ctx.timer_start()
    for _ in range(50):
        ctx.ranksum_dev(d_ps, g1, g2, out)
    res.append(ctx.timer_stop() / 50)
print(os.path.relpath(_ffi.LIB_PATH), " ".join(f"{x:.4f}" for x in res), "ms per 1M rows (whole call)", flush=True)
