import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splicedice_amd import synth
from splicedice_amd.engine import Context
ctx = Context(0)
n, s = int(sys.argv[1]), int(sys.argv[2])
ps = synth.make_ps_matrix(n, s, 3)
d_ps = ctx.to_device(ps)
h = s // 2
g1, g2 = ctx.to_device(np.arange(0, h, dtype=np.int32)), ctx.to_device(np.arange(h, s, dtype=np.int32))
out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64), med1=ctx.empty(n, np.float32), med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32), mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
if len(sys.argv) > 3: ctx.set_param("ranksum.variant", int(sys.argv[3]))
ctx.prof_enable(1)
for _ in range(2): ctx.ranksum_dev(d_ps, g1, g2, out)
ctx.prof_reset()
for _ in range(3): ctx.ranksum_dev(d_ps, g1, g2, out)
print(n, s, {k: (v[0], round(v[1] / v[0], 3)) for k, v in ctx.prof_report().items()}, "GB/s alg", flush=True)
