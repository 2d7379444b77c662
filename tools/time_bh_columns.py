import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splicedice_amd.engine import Context
ctx = Context(0)
n, cols = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
p = rng.random((n, cols))
d = ctx.to_device(p)
for _ in range(2):
    ctx.timer_start(); ctx.bh_columns_dev(d); ms = ctx.timer_stop()
    print(f"bh_columns_dev {n} x {cols}: {ms:.2f} ms  ({n*cols/ms/1e6:.2f} G p-values/s)", flush=True)
