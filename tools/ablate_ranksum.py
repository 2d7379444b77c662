import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splicedice_amd import synth
from splicedice_amd.engine import Context
ctx = Context(0)
n, s = 1000000, 100
ps = synth.make_ps_matrix(n, s, 3)
d_ps = ctx.to_device(ps)
g1 = ctx.to_device(np.arange(0, 50, dtype=np.int32)); g2 = ctx.to_device(np.arange(50, 100, dtype=np.int32))
out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64), med1=ctx.empty(n, np.float32), med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32), mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
ctx.prof_enable(1)
ctx.set_param("ranksum.variant", 1)     # the lane kernel (the default for these sizes is the pair kernel)
for ab in (0, 1, 2, 3):
    ctx.set_param("ranksum.ablate", ab)
    for _ in range(2): ctx.ranksum_dev(d_ps, g1, g2, out)
    ctx.prof_reset()
    for _ in range(5): ctx.ranksum_dev(d_ps, g1, g2, out)
    k, ms = ctx.prof_query("ranksum_lane_kernel")
    print("ablate", ab, "ms", ms / k, flush=True)
