#!/usr/bin/env python3
"""Same-process timing of sdice_fisher_pairs_dev on the bench's data: ab_fisher.py n s [param=value,...] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
n, s = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:] or [""]
DEFAULTS = {"fisher.refill": 12, "fisher.unroll": 16, "fisher.count_steps": 0}
ctx = Context(0)
junc = synth.make_junctions(n, 4)
counts_in = synth.make_counts(n, s, 40)
row_of, row_ptr, col = ctx.cluster(*junc)
counts = np.zeros_like(counts_in); counts[row_of] = counts_in
d_counts, d_rp, d_col = ctx.to_device(counts), ctx.to_device(row_ptr), ctx.to_device(col)
d_excl = ctx.empty((n, s), np.int64)
pairs = s * (s - 1) // 2
d_p = ctx.empty((n, pairs), np.float64)
ctx.ps_dev(d_counts, d_rp, d_col, d_excl, None)
ref = None
for rep in range(2):
    for c in cfgs:
        kv = [x.split("=") for x in c.split(",") if x]
        for k, v in kv:
            ctx.set_param(k, int(v))
        ms = []
        for it in range(3):
            ctx.sync(); ctx.timer_start(); ctx.fisher_pairs_dev(d_counts, d_excl, d_p); ms.append(ctx.timer_stop())
        extra = ""
        if any(k == "fisher.count_steps" and int(v) for k, v in kv):
            u, t = ctx.fisher_step_stats()
            extra = f"  useful lane-steps {u:.3e} of {t:.3e} issued = {u / max(t, 1):.3f}; {u / (n * pairs):.1f} per pair"
        for k, v in kv:
            ctx.set_param(k, DEFAULTS[k])
        got = d_p.offset(0, (4, pairs)).to_host()
        if ref is None:
            ref = got
        err = np.max(np.abs(got - ref) / ref)
        print(f"rep {rep} [{c}] {min(ms[1:]):.3f} ms  ({n * pairs / min(ms[1:]) / 1e6:.2f} G p/s)  max rel diff to first config {err:.1e}{extra}", flush=True)
