#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (numbers quoted in DESIGN.md section 6; never bench `value`)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splicedice_amd import synth
from splicedice_amd.engine import Context
ctx = Context(0)
n, s = 1_000_000, 100
junc = synth.make_junctions(n, 2)
counts = synth.make_counts(n, s, 20)
row_of, rp, col = ctx.cluster(*junc)
for _ in range(2):
    t = time.perf_counter(); ps = ctx.ps(counts, rp, col); dt = time.perf_counter() - t
print("sdice_ps host buffers: %.1f ms -> %.2e entries/s (%.1f GB/s over the link)" % (dt * 1e3, n * s / dt, 8 * n * s / dt / 1e9))
g1, g2 = np.arange(0, 50, dtype=np.int32), np.arange(50, 100, dtype=np.int32)
psq = synth.make_ps_matrix(n, s, 3)
for _ in range(2):
    t = time.perf_counter(); r = ctx.ranksum(psq, g1, g2); dt = time.perf_counter() - t
print("sdice_ranksum host buffers: %.1f ms -> %.2e rows/s" % (dt * 1e3, n / dt))
t = time.perf_counter(); ctx.cluster(*junc); dt = time.perf_counter() - t
print("sdice_cluster host buffers: %.1f ms -> %.2e junctions/s" % (dt * 1e3, n / dt))
