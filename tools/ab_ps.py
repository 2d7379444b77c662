#!/usr/bin/env python3
"""Same-process interleaved A/B of PS launch configurations.
  ab_ps.py n s cfg [cfg ...]      cfg = comma-separated ps.* parameters, e.g. dma=1,threads=512,lds_bytes=40960,halo_rows=16
Every configuration is checked once against the first one (bit-equal PS), then timed in 4 interleaved rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
n, s = int(sys.argv[1]), int(sys.argv[2])
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[3:]]
keys = sorted({k for c in cfgs for k in c})
DEFAULTS = {"halo_rows": -1, "tile_rows": 0, "chunk_cols": 0, "gen1": 0, "use_reach": 1, "nt_loads": 1, "prio": 1, "xcd_remap": 1, "quantize3": 0}
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

def apply(c):
    for k in keys:
        if k in c:
            ctx.set_param("ps." + k, c[k])
        elif k in DEFAULTS:
            ctx.set_param("ps." + k, DEFAULTS[k])
        else:
            raise SystemExit(f"every configuration must give ps.{k} (no default known here)")

ref = None
m = min(n, 300_000)
for c in cfgs:
    apply(c)
    ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
    ctx.sync()
    got = d_ps.offset(0, (m, s)).to_host().view(np.uint32)
    tail = d_ps.offset((n - 1000) * s, (1000, s)).to_host().view(np.uint32)
    if ref is None:
        ref = (got, tail)
    else:
        same = np.array_equal(got, ref[0]) and np.array_equal(tail, ref[1])
        print(f"cfg {c}: {'bit-equal to the first configuration' if same else 'DIFFERS from the first configuration'}", flush=True)
for rep in range(4):
    for c in cfgs:
        apply(c)
        for _ in range(5):
            ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
        ctx.sync()
        ctx.timer_start()
        for _ in range(40):
            ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
        ms = ctx.timer_stop() / 40
        print(f"rep {rep} {c}: {ms:.4f} ms  {n * s * 8 / ms / 1e6:.0f} GB/s", flush=True)
