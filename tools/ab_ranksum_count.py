#!/usr/bin/env python3
"""Same-process A/B of ranksum_count_kernel (groups of 500 v 500 on 3-decimal PS values): ab_ranksum_count.py [rows]
ranksum.ablate = 1 switches the NaN-free fast path of the compaction off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
n, s = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 625_000, 1000
ctx = Context(0)
blk = 25_000
ps = synth.make_ps_matrix(blk, s, 3)
if '--keep-nan' not in sys.argv:
    ps[np.isnan(ps)] = 0.5           # (quant output at the bench's counts: a NaN in ~0.3 % of the rows)
    ps[::400, 7] = np.nan
d_ps = ctx.empty((n, s), np.float32)
for a in range(0, n, blk):
    b = min(n, a + blk)
    d_ps.offset(a * s, (b - a, s)).upload(ps[: b - a])
g1, g2 = ctx.to_device(np.arange(0, 500, dtype=np.int32)), ctx.to_device(np.arange(500, 1000, dtype=np.int32))
out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64), med1=ctx.empty(n, np.float32),
           med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32), mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
print("rows with a NaN among the selected values:", float(np.isnan(ps).any(axis=1).mean()))
for rep in range(3):
    for abl in (0, 1):
        ctx.set_param("ranksum.ablate", abl)
        for _ in range(3):
            ctx.ranksum_dev(d_ps, g1, g2, out)
        ctx.sync(); ctx.timer_start()
        for _ in range(10):
            ctx.ranksum_dev(d_ps, g1, g2, out)
        print(f"rep {rep} ranksum.ablate={abl}: {ctx.timer_stop() / 10:.4f} ms per {n} rows (whole call)", flush=True)
ctx.set_param("ranksum.ablate", 0)
