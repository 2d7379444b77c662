#!/usr/bin/env python3
"""Same-process A/B of the fast clustering chain: ab_cluster.py n [cluster.param=value,...] ...  (per-kernel us via the
library's profiling switch, asynchronous calls)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
DEFAULTS = {"cluster.bucket_mean": 2048, "cluster.spb": 0, "cluster.sample_sort": 1, "cluster.nb_grid": 0}
n = int(sys.argv[1])
cfgs = sys.argv[2:] or [""]
ctx = Context(0)
junc = synth.make_junctions(n, 2)
d = [ctx.to_device(x) for x in junc]
d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
ctx.cluster_dev(*d, d_row_of, d_rp)
for rep in range(2):
    for c in cfgs:
        kv = [x.split("=") for x in c.split(",") if x]
        for k, v in kv:
            ctx.set_param(k, int(v))
        ctx.prof_enable(0)
        for _ in range(3):
            ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
        ctx.sync()
        ctx.timer_start()
        for _ in range(20):
            ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
        wall = ctx.timer_stop() / 20
        ctx.sync()
        ctx.prof_enable(1); ctx.prof_reset()
        for _ in range(5):
            ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
        ctx.sync()
        r = {k.replace("_kernel", ""): round(v[1] / 5 * 1000, 1) for k, v in ctx.prof_report().items()}
        ctx.prof_enable(0)
        for k, v in kv:
            ctx.set_param(k, DEFAULTS[k])
        print(f"rep {rep} [{c}] async {wall:.4f} ms  {json.dumps(r)}", flush=True)
