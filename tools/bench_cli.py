#!/usr/bin/env python3
"""End-to-end `quant` -> `compare_sample_sets` through the command-line entry points on synthetic
junction files (stage times from the reference-style banners).  Not part of the product."""
import argparse, os, sys, tempfile, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import synth
from splicedice_amd.__main__ import main

ap = argparse.ArgumentParser()
ap.add_argument("--junctions", type=int, default=200000)
ap.add_argument("--samples", type=int, default=40)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="sdice_cli_")
t = time.time()
cr, left, right, strand = synth.make_junctions(a.junctions, 9, n_chrom=24)
names = synth.chrom_names(24)
rng = np.random.default_rng(9)
with open(os.path.join(d, "manifest.tsv"), "w") as mf:
    for s in range(a.samples):
        cnt = synth.make_counts(a.junctions, 1, 900 + s, zero_frac=0.15)[:, 0]
        keep = np.flatnonzero(rng.random(a.junctions) > 0.1)
        path = os.path.join(d, f"s{s}.junc.bed")
        with open(path, "w") as fh:
            fh.write("".join(f"{names[cr[j]]}\t{left[j]}\t{right[j]}\te:1.50:1.20;o:20;m:GT_AG;a:?\t{cnt[j]}\t{'+-'[strand[j]]}\n" for j in keep))
        mf.write(f"s{s}\t{path}\tm\t{'A' if s < a.samples // 2 else 'B'}\n")
with open(os.path.join(d, "g1.tsv"), "w") as f1, open(os.path.join(d, "g2.tsv"), "w") as f2:
    for s in range(a.samples):
        (f1 if s < a.samples // 2 else f2).write(f"s{s}\tp\tm\tc\n")
print(f"generated {a.samples} files x ~{int(a.junctions * 0.9)} lines in {time.time() - t:.1f}s", flush=True)
t = time.time()
main(["quant", "-m", os.path.join(d, "manifest.tsv"), "-o", os.path.join(d, "out")])
tq = time.time() - t
t = time.time()
main(["compare_sample_sets", "--psiSPLICEDICE", os.path.join(d, "out_allPS.tsv"), "-m1", os.path.join(d, "g1.tsv"),
      "-m2", os.path.join(d, "g2.tsv"), "-o", os.path.join(d, "cmp.tsv")])
tc = time.time() - t
print(f"quant total {tq:.2f}s   compare_sample_sets total {tc:.2f}s   PS table {os.path.getsize(os.path.join(d, 'out_allPS.tsv')) / 1e6:.0f} MB")
