#!/usr/bin/env python3
"""Per-kernel times of BH over one masked vector (sdice_bh_masked_dev): ab_bh_vector.py m"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd.engine import Context
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = Context(0)
rng = np.random.default_rng(5)
p = rng.random(m) ** 2
p[rng.random(m) < 0.1] = 1.0
tested = (rng.random(m) < 0.95).astype(np.uint8)
d_p, d_t, d_q = ctx.to_device(p), ctx.to_device(tested), ctx.empty(m, np.float64)
for rep in range(2):
    for path, mean in ((1, 0), (2, 0)):
        ctx.set_param('bh.vector_path', path)
        ctx.prof_enable(0)
        for _ in range(3):
            ctx.bh_masked_dev(d_p, d_t, d_q)
        ctx.sync(); ctx.timer_start()
        for _ in range(20):
            ctx.bh_masked_dev(d_p, d_t, d_q)
        ms = ctx.timer_stop() / 20
        ctx.prof_enable(1); ctx.prof_reset()
        for _ in range(5):
            ctx.bh_masked_dev(d_p, d_t, d_q)
        ctx.sync()
        r = {k.replace("_kernel", ""): round(v[1] / 5 * 1000, 1) for k, v in ctx.prof_report().items()}
        ctx.prof_enable(0)
        print(f"rep {rep} path {path} (1 = radix, 2 = sample sort): {ms:.4f} ms  {json.dumps(r)}", flush=True)
