#!/usr/bin/env python3
"""Feasibility probe: does the f64-VALU-bound Fisher kernel overlap with the memory / LDS-bound column BH when they run on
two streams (two contexts) of one GPU?  Times Fisher alone, BH alone, and both enqueued back to back on their own streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
n, s = 25000, 200
A, B = Context(0), Context(0)
junc = synth.make_junctions(n, 4)
counts_in = synth.make_counts(n, s, 40)
row_of, row_ptr, col = A.cluster(*junc)
counts = np.zeros_like(counts_in); counts[row_of] = counts_in
pairs = s * (s - 1) // 2
dA_counts, dA_rp, dA_col = A.to_device(counts), A.to_device(row_ptr), A.to_device(col)
dA_excl = A.empty((n, s), np.int64)
dA_p = A.empty((n, pairs), np.float64)
A.ps_dev(dA_counts, dA_rp, dA_col, dA_excl, None)
A.fisher_pairs_dev(dA_counts, dA_excl, dA_p); A.sync()
# an independent p-value table for the BH context
rng = np.random.default_rng(1)
blk = rng.random((1000, pairs)) ** 2
dB_src = B.empty((n, pairs), np.float64)
for a in range(0, n, 1000):
    dB_src.offset(a * pairs, (1000, pairs)).upload(blk)
dB = B.empty((n, pairs), np.float64)
def bh():
    B.copy2d_dev(dB.ptr, pairs * 8, dB_src.ptr, pairs * 8, pairs * 8, n)
    B.bh_columns_dev(dB)
for rep in range(3):
    A.sync(); B.sync(); t = time.perf_counter(); A.fisher_pairs_dev(dA_counts, dA_excl, dA_p); A.sync(); tf = time.perf_counter() - t
    A.sync(); B.sync(); t = time.perf_counter(); bh(); B.sync(); tb = time.perf_counter() - t
    A.sync(); B.sync(); t = time.perf_counter(); A.fisher_pairs_dev(dA_counts, dA_excl, dA_p); bh(); A.sync(); B.sync(); both = time.perf_counter() - t
    print(f"rep {rep}: Fisher {tf * 1e3:.2f} ms, copy + BH {tb * 1e3:.2f} ms, both on two streams {both * 1e3:.2f} ms (sum {1e3 * (tf + tb):.2f})", flush=True)
