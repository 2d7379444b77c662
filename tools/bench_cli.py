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

# ---- `pairwise` through the CLI at config-4 width (200 samples = 19 900 pair columns); the p-value matrix
# stays in HBM and leaves in row slabs, so the host's peak memory is independent of the number of junctions
import resource
from splicedice_amd import textio
from splicedice_amd.engine import Context
npw = int(os.environ.get("SDICE_PW_JUNCTIONS", "5000"))
s = 200
cr, left, right, strand = synth.make_junctions(npw, 11, n_chrom=24)
with Context(0) as c:
    row_of, row_ptr, col = c.cluster(cr, left, right, strand)
order = np.argsort(row_of)
pw_names = [f"{names[cr[j]]}:{left[j]}-{right[j]}:{'+-'[strand[j]]}" for j in order]
counts = synth.make_counts(npw, s, 77)
hdr = "cluster\t" + "\t".join(f"s{k}" for k in range(s)) + "\n"
textio.write_table(os.path.join(d, "pw_inclusionCounts.tsv"), hdr, pw_names, counts, ".0f")
textio.write_clusters(os.path.join(d, "pw_allClusters.tsv"), pw_names, row_ptr, col)
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
t = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    main(["pairwise", "--inclusionSPLICEDICE", os.path.join(d, "pw_inclusionCounts.tsv"), "-c", os.path.join(d, "pw_allClusters.tsv"),
          "-o", os.path.join(d, "pairwise.tsv")])
tp = time.time() - t
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
size = os.path.getsize(os.path.join(d, "pairwise.tsv"))
print(f"pairwise {npw} junctions x {s} samples ({s * (s - 1) // 2} pair columns, BH per column): {tp:.2f}s, "
      f"{npw * s * (s - 1) // 2 / tp:.3e} p-values/s through the CLI, output {size / 1e9:.2f} GB, "
      f"peak RSS {rss1 / 1e6:.2f} GB (before: {rss0 / 1e6:.2f} GB); matrix in HBM: {npw * s * (s - 1) // 2 * 8 / 1e9:.2f} GB")
import shutil
shutil.rmtree(d, ignore_errors=True)
