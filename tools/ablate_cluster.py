#!/usr/bin/env python3
"""Per-kernel timing of the fast clustering chain under the cluster.ablate switches (tuning aid; the switches
exist only in a library built with `make -C splicedice_amd/csrc EXTRA=-DSDICE_CLUSTER_ABLATE=1`).
Results are wrong under ablation; calls are asynchronous and their status is discarded."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context, SdiceError

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
masks = [int(x) for x in sys.argv[2:]] or [0]
ctx = Context(0)
junc = synth.make_junctions(n, 2)
d = [ctx.to_device(x) for x in junc]
d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
ctx.cluster_dev(*d, d_row_of, d_rp)
for m in masks:
    ctx.set_param("cluster.sample_sort", 0 if m < 0 else 1)      # negative mask: the workgroup-wide network instead
    tag = "network" if m < 0 else "sample-sort"
    m = abs(m) if m != -9999 else 0
    ctx.set_param("cluster.ablate", m)
    ctx.prof_enable(0)
    for _ in range(3):
        ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
    try:
        ctx.sync()
    except SdiceError:
        pass
    ctx.timer_start()
    for _ in range(10):
        ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
    wall = ctx.timer_stop() / 10
    try:
        ctx.sync()
    except SdiceError:
        pass
    ctx.prof_enable(1)
    ctx.prof_reset()
    for _ in range(5):
        ctx.cluster_dev(*d, d_row_of, d_rp, sync=False)
    try:
        ctx.sync()
    except SdiceError:
        pass
    rep = {k: round(v[1] / 5 * 1000, 1) for k, v in ctx.prof_report().items()}
    print(tag, "ablate", m, "async %.3f ms;" % wall, json.dumps(rep), "sum %.1f us" % sum(rep.values()), flush=True)
ctx.set_param("cluster.ablate", 0)
ctx.set_param("cluster.sample_sort", 1)
