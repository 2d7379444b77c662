#!/usr/bin/env python3
"""Time the rank-sum call with the library given on the command line (A/B of two builds inside ONE gpurun call, i.e. on
one GPU): `python tools/time_ranksum_lib.py build/base_lib/libsplicedice_hip.so [rows samples]` -- 1 M rows of 50 v 50 by
default, `625000 1000` is the config-5 shard (500 v 500, counting kernel)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import _ffi, synth
if len(sys.argv) > 1:
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
from splicedice_amd.engine import Context
n, s = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1_000_000, 100)
h = s // 2
blk = min(n, 200_000 if s <= 100 else 25_000)
n -= n % blk
ctx = Context(0)
ps = synth.make_ps_matrix(blk, s, 3)
d_ps = ctx.empty((n, s), np.float32)
for a in range(0, n, blk):
    d_ps.offset(a * s, (blk, s)).upload(ps)
g1, g2 = ctx.to_device(np.arange(0, h, dtype=np.int32)), ctx.to_device(np.arange(h, 2 * h, dtype=np.int32))
out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64), med1=ctx.empty(n, np.float32),
           med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32), mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
reps = 50 if s <= 100 else 10
for _ in range(reps // 2):
    ctx.ranksum_dev(d_ps, g1, g2, out)
ctx.sync()
res = []
for rep in range(3):
    ctx.timer_start()
    for _ in range(reps):
        ctx.ranksum_dev(d_ps, g1, g2, out)
    res.append(ctx.timer_stop() / reps)
print(os.path.relpath(_ffi.LIB_PATH), " ".join(f"{x:.4f}" for x in res), f"ms per {n} rows x {s} (whole call)", flush=True)
