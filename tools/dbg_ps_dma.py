#!/usr/bin/env python3
"""debug aid: DMA tile kernel against the register-staged one on small shapes (host API, foreign CSR)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.engine import Context
import oracle_np as O
ctx = Context(0)
KERN = int(os.environ.get("PS_KERN", "0"))
shapes = [(3000, 100, 1), (5000, 4, 2), (2000, 8, 3), (2000, 16, 4), (1000, 64, 5), (300, 1000, 4), (1500, 260, 6)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for n, s, seed in shapes:
    cr, left, right, strand = synth.make_junctions(n, seed, n_chrom=4)
    _, row_ptr, col = O.cluster_csr(cr, left, right, strand)
    counts = synth.make_counts(n, s, seed + 50)
    ctx.set_param("ps.dma", 2)
    ps0, ex0 = ctx.ps(counts, row_ptr, col, want_excl=True)
    ctx.set_param("ps.dma", KERN)
    ps1, ex1 = ctx.ps(counts, row_ptr, col, want_excl=True)
    bad = np.flatnonzero((ex0 != ex1).any(axis=1))
    print(f"n {n} s {s}: {bad.size} rows differ", flush=True)
    if bad.size:
        print("   first", bad[:20], "last", bad[-5:])
        r = bad[0]
        print("   row", r, "deg", row_ptr[r + 1] - row_ptr[r], "col", col[row_ptr[r]:row_ptr[r + 1]], "want", ex0[r][:4], "got", ex1[r][:4])
        runs = np.split(bad, np.flatnonzero(np.diff(bad) > 1) + 1)
        print("   runs:", [(int(x[0]), int(x[-1])) for x in runs[:12]])

# device path with the clustering's reach words
for n, s, seed in [(20000, 100, 11), (50000, 36, 12), (30000, 500, 13), (200000, 100, 14)]:
    cr, left, right, strand = synth.make_junctions(n, seed)
    d = [ctx.to_device(x) for x in (cr, left, right, strand)]
    d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_col, nnz = ctx.cluster_dev(*d, d_row_of, d_rp)
    counts = synth.make_counts(n, s, seed + 50)
    d_counts, d_ps = ctx.to_device(counts), ctx.empty((n, s), np.float32)
    out = []
    for dma in (2, KERN):
        ctx.set_param("ps.dma", dma)
        d_ps.memset(0xff)
        ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
        ctx.sync()
        out.append(d_ps.to_host().view(np.uint32))
    bad = np.flatnonzero((out[0] != out[1]).any(axis=1))
    print(f"device path n {n} s {s} nnz {nnz}: {bad.size} rows differ", flush=True)
    if bad.size:
        print("   first", bad[:20], "last", bad[-5:])
