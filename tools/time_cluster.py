#!/usr/bin/env python3
"""Per-kernel timing of sdice_cluster_dev for a few parameter settings (tuning aid, not product)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
settings = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
ctx = Context(0)
junc = synth.make_junctions(n, 2)
d = [ctx.to_device(x) for x in junc]
d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
for st in settings:
    for k, v in st.items():
        ctx.set_param(k, int(v))
    ctx.prof_enable(0)
    for _ in range(3):
        ctx.cluster_dev(*d, d_row_of, d_rp)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        ctx.cluster_dev(*d, d_row_of, d_rp)
    wall = ctx.timer_stop() / 10
    ctx.timer_start()
    for _ in range(10):
        ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
    wall_async = ctx.timer_stop() / 10
    ctx.sync()
    ctx.prof_enable(1)
    ctx.prof_reset()
    for _ in range(5):
        ctx.cluster_dev(*d, d_row_of, d_rp)
    rep = {k: round(v[1] / 5 * 1000, 1) for k, v in ctx.prof_report().items()}
    print(st, "async %.3f ms;" % wall_async, "cluster_dev %.3f ms (unprofiled); kernels us/call: %s; sum %.1f us" % (wall, json.dumps(rep), sum(rep.values())),
          flush=True)
    for k in st:
        ctx.set_param(k, 0)
