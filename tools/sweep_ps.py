#!/usr/bin/env python3
"""Tuning sweep of the K3 PS kernel launch parameters on one GPU (not part of the product)."""
import argparse, itertools, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--s", type=int, default=100)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--lds", type=str, default="32768,65536,81920,163840")
ap.add_argument("--threads", type=str, default="256,512,1024")
ap.add_argument("--remap", type=str, default="1,0")
ap.add_argument("--chunk", type=str, default="0")
ap.add_argument("--deg0", action="store_true")
ap.add_argument("--ablate", type=str, default="0")
ap.add_argument("--tile_rows", type=str, default="0")
ap.add_argument("--halo", type=str, default="-1")
a = ap.parse_args()
ctx = Context(0)
print(ctx.device_info(), flush=True)
t = time.time()
cr, l, r, st = synth.make_junctions(a.n, 2)
rng = np.random.default_rng(3)
counts = rng.integers(0, 200, size=(a.n, a.s), dtype=np.int32)
print("gen %.1fs" % (time.time() - t), flush=True)
d = [ctx.to_device(x) for x in (cr, l, r, st)]
d_row_of, d_rp = ctx.empty(a.n, np.int32), ctx.empty(a.n + 1, np.int64)
ctx.prof_enable(True)
for _ in range(3):
    ctx.prof_reset(); ctx.timer_start()
    d_col, nnz = ctx.cluster_dev(*d, d_row_of, d_rp)
    ms = ctx.timer_stop()
print("cluster_dev %.3f ms nnz=%d deg=%.2f" % (ms, nnz, nnz / a.n), {k: round(v[1], 3) for k, v in ctx.prof_report().items()}, flush=True)
d_counts, d_ps = ctx.to_device(counts), ctx.empty((a.n, a.s), np.float32)
if a.deg0:
    d_rp = ctx.to_device(np.zeros(a.n + 1, np.int64))
res = []
for lds, th, remap, chunk, tr, halo, ab in itertools.product(*[[int(x) for x in v.split(",")] for v in (a.lds, a.threads, a.remap, a.chunk, a.tile_rows, a.halo, a.ablate)]):
    ctx.set_param("ps.ablate", ab)
    ctx.set_param("ps.tile_rows", tr); ctx.set_param("ps.halo_rows", halo)
    ctx.set_param("ps.lds_bytes", lds); ctx.set_param("ps.threads", th); ctx.set_param("ps.xcd_remap", remap)
    ctx.set_param("ps.chunk_cols", chunk)
    for _ in range(2):
        ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
    ctx.prof_reset()
    for _ in range(a.iters):
        ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
    k, ms = ctx.prof_query("ps_tile_v3_kernel")
    ms /= k
    gbs = a.n * a.s * 8 / ms / 1e6
    res.append((gbs, lds, th, remap, chunk, ms))
    print("ab=%d tr=%d halo=%d " % (ab, tr, halo), end="");print("lds=%6d threads=%4d remap=%d chunk=%3d  %.4f ms  %.1f GB/s (alg)  %.3e entries/s" % (lds, th, remap, chunk, ms, gbs, a.n * a.s / ms * 1e3), flush=True)
res.sort(reverse=True)
print("best", res[0])
