#!/usr/bin/env python3
"""Same-process A/B of the whole quant step (asynchronous clustering + PS): ab_quant_step.py n s [param=value,...] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
n, s = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:] or [""]
DEFAULTS = {"cluster.spb": 0, "cluster.bucket_mean": 2048, "cluster.sample_sort": 1, "ps.prio": 1, "ps.nt_loads": 1, "ps.halo_rows": -1,
            "ps.tile_rows": 0, "ps.lds_bytes": 81920, "ps.threads": 1024}
ctx = Context(0)
junc = synth.make_junctions(n, 2)
d = [ctx.to_device(x) for x in junc]
d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
d_col, nnz = ctx.cluster_dev(*d, d_row_of, d_rp)
blk = synth.make_counts(min(n, 200_000), s, 20)
d_counts, d_ps = ctx.empty((n, s), np.int32), ctx.empty((n, s), np.float32)
for a in range(0, n, blk.shape[0]):
    b = min(n, a + blk.shape[0]); d_counts.offset(a * s, (b - a, s)).upload(blk[: b - a])
def step():
    ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
    ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
for _ in range(50): step()
ctx.sync()
for rep in range(3):
    for c in cfgs:
        kv = [x.split("=") for x in c.split(",") if x]
        for k, v in kv: ctx.set_param(k, int(v))
        for _ in range(20): step()
        ctx.sync(); ctx.timer_start()
        for _ in range(200): step()
        ms = ctx.timer_stop() / 200
        ctx.sync()
        for k, v in kv: ctx.set_param(k, DEFAULTS.get(k, 0))
        print(f"rep {rep} [{c}] {ms:.4f} ms per step  {n * s / ms / 1e6:.1f} G entries/s", flush=True)
