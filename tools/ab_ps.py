#!/usr/bin/env python3
"""Same-process interleaved A/B of PS launch configurations: ab_ps.py n s lds:threads[:bits] ... (bits: 1 = load-phase priority, 2 = non-temporal window loads; default 3)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
n, s = int(sys.argv[1]), int(sys.argv[2])
cfgs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[3:]]
ctx = Context(0)
cr, l, r, st = synth.make_junctions(n, 2)
d = [ctx.to_device(x) for x in (cr, l, r, st)]
d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
d_col, nnz = ctx.cluster_dev(*d, d_row_of, d_rp)
blk = synth.make_counts(min(n, 200_000), s, 20)
d_counts, d_ps = ctx.empty((n, s), np.int32), ctx.empty((n, s), np.float32)
for a in range(0, n, blk.shape[0]):
    b = min(n, a + blk.shape[0])
    d_counts.offset(a * s, (b - a, s)).upload(blk[: b - a])
for _ in range(10):
    ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
ctx.sync()
for rep in range(4):
    for c in cfgs:
        ctx.set_param("ps.lds_bytes", c[0]); ctx.set_param("ps.threads", c[1]); ctx.set_param("ps.halo_rows", -1); ctx.set_param("ps.prio", (c[2] & 1) if len(c) > 2 else 1); ctx.set_param("ps.nt_loads", ((c[2] >> 1) & 1) if len(c) > 2 else 1)
        for _ in range(3):
            ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
        ctx.sync()
        ctx.timer_start()
        for _ in range(30):
            ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
        ms = ctx.timer_stop() / 30
        print(f"rep {rep} lds={c[0]} threads={c[1]} bits={c[2] if len(c) > 2 else 3}: {ms:.4f} ms  {n * s * 8 / ms / 1e6:.0f} GB/s", flush=True)
