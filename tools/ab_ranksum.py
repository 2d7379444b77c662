#!/usr/bin/env python3
"""Same-process A/B of the rank-sum variants for groups <= 64 (boxes differ by ~15 % between runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
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
for rep in range(3):
    for variant, name in ((4, "float lane-pair"), (0, "16-bit keys (auto)"), (1, "lane")):
        ctx.set_param("ranksum.variant", variant)
        for _ in range(3):
            ctx.ranksum_dev(d_ps, g1, g2, out)
        ctx.sync()
        ctx.timer_start()
        for _ in range(20):
            ctx.ranksum_dev(d_ps, g1, g2, out)
        ms = ctx.timer_stop() / 20
        print(f"rep {rep} variant {variant} ({name}): {ms:.4f} ms per 1M rows (whole call incl. finish kernel)", flush=True)
ctx.set_param("ranksum.variant", 0)
