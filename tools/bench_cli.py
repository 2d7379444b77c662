#!/usr/bin/env python3
"""End-to-end runs of the sub-commands through the command-line entry points at BASELINE scale, with a stage breakdown
(parse, H2D, kernels, D2H, format, write) -> one JSON document (profiles/rNN_cli.json).  Not part of the product.

    python tools/bench_cli.py --out gpurun_out/r4/cli.json [--quant-junctions 1000000 --quant-samples 100
                              --pairwise-junctions 25000 --pairwise-samples 200]

quant: a manifest of `--quant-samples` synthetic .junc.bed files over one junction set (the reference's input format,
SPLICEDICE.py:147-228); compare_sample_sets: the _allPS.tsv that quant wrote, two halves of the samples;
pairwise: an _inclusionCounts.tsv / _allClusters.tsv pair written by the library's own writers.
Stage seconds come from splicedice_amd/_stages.py (SDICE_STAGES=1) and the table writers' own format / write split.
"""
import argparse, contextlib, io, json, os, resource, shutil, sys, tempfile, time
os.environ["SDICE_STAGES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from splicedice_amd import _stages, synth, textio
from splicedice_amd.__main__ import main
from splicedice_amd.engine import Context

ap = argparse.ArgumentParser()
ap.add_argument("--quant-junctions", type=int, default=1_000_000)
ap.add_argument("--quant-samples", type=int, default=100)
ap.add_argument("--pairwise-junctions", type=int, default=25_000)
ap.add_argument("--pairwise-samples", type=int, default=200)
ap.add_argument("--out", default="")
ap.add_argument("--tmp", default=None)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="sdice_cli_", dir=a.tmp)
doc = {"tmp": d, "host_cpus": len(os.sched_getaffinity(0))}


def run(argv):
    _stages.take()
    t = time.perf_counter()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        main(argv)
    wall = time.perf_counter() - t
    st = _stages.take()
    st["wall_s"] = wall
    st["other_s"] = wall - sum(v for k, v in st.items() if k in ("parse", "cluster", "h2d", "kernels", "d2h", "format+write"))
    return st


try:
    # ------------------------------------------------------------------ inputs of quant
    nj, ns = a.quant_junctions, a.quant_samples
    t = time.time()
    cr, left, right, strand = synth.make_junctions(nj, 9, n_chrom=24)
    names = synth.chrom_names(24)
    rng = np.random.default_rng(9)
    chrom_col = np.array(names)[cr]
    strand_col = np.array(["+", "-"])[strand]
    # the lines of a file in the order bam_to_junc_bed writes them (`for junction in sorted(counts)`, bam_to_junc_bed.py:162:
    # chromosome as a string, left, right, strand)
    by_key = np.lexsort((strand_col, right, left, chrom_col))
    rank_of = np.empty(nj, np.int64)
    rank_of[by_key] = np.arange(nj)
    in_bytes = 0
    with open(os.path.join(d, "manifest.tsv"), "w") as mf:
        for s in range(ns):
            cnt = synth.make_counts(nj, 1, 900 + s, zero_frac=0.15)[:, 0]
            keep = np.flatnonzero(rng.random(nj) > 0.1)
            keep = keep[np.argsort(rank_of[keep], kind="stable")]
            path = os.path.join(d, f"s{s}.junc.bed")
            # (vectorised text assembly: generation is not what is measured)
            cols = [chrom_col[keep], left[keep].astype(str), right[keep].astype(str),
                    np.full(keep.size, "e:1.50:1.20;o:20;m:GT_AG;a:?"), cnt[keep].astype(str), strand_col[keep]]
            lines = cols[0]
            for c in cols[1:]:
                lines = np.char.add(np.char.add(lines, "\t"), c)
            with open(path, "w") as fh:
                fh.write("\n".join(lines.tolist()) + "\n")
            in_bytes += os.path.getsize(path)
            mf.write(f"s{s}\t{path}\tm\t{'A' if s < ns // 2 else 'B'}\n")
    with open(os.path.join(d, "g1.tsv"), "w") as f1, open(os.path.join(d, "g2.tsv"), "w") as f2:
        for s in range(ns):
            (f1 if s < ns // 2 else f2).write(f"s{s}\tp\tm\tc\n")
    doc["generation_s"] = round(time.time() - t, 1)
    print(f"generated {ns} files, {in_bytes / 1e9:.2f} GB in {doc['generation_s']} s", flush=True)

    # ------------------------------------------------------------------ quant, compare_sample_sets
    st = run(["quant", "-m", os.path.join(d, "manifest.tsv"), "-o", os.path.join(d, "out")])
    ps_bytes = os.path.getsize(os.path.join(d, "out_allPS.tsv"))
    n_out = sum(1 for _ in open(os.path.join(d, "out_junctions.bed")))
    st.update(junctions_in_files=nj, junction_rows_out=n_out, samples=ns, input_bytes=in_bytes, allPS_bytes=ps_bytes,
              inclusionCounts_bytes=os.path.getsize(os.path.join(d, "out_inclusionCounts.tsv")),
              entries_per_s_through_cli=n_out * ns / st["wall_s"],
              values_formatted_per_s=2 * n_out * ns / max(st["writer_format_s"], 1e-9))
    doc["quant"] = st
    print("quant", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)

    st = run(["compare_sample_sets", "--psiSPLICEDICE", os.path.join(d, "out_allPS.tsv"), "-m1", os.path.join(d, "g1.tsv"),
              "-m2", os.path.join(d, "g2.tsv"), "-o", os.path.join(d, "cmp.tsv")])
    st.update(rows=n_out, samples=ns, rows_per_s_through_cli=n_out / st["wall_s"], output_bytes=os.path.getsize(os.path.join(d, "cmp.tsv")))
    doc["compare_sample_sets"] = st
    print("compare_sample_sets", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
    for f in os.listdir(d):
        if f.endswith(".junc.bed"):
            os.remove(os.path.join(d, f))

    # ------------------------------------------------------------------ pairwise at config-4 width
    npw, s = a.pairwise_junctions, a.pairwise_samples
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
    st = run(["pairwise", "--inclusionSPLICEDICE", os.path.join(d, "pw_inclusionCounts.tsv"), "-c", os.path.join(d, "pw_allClusters.tsv"),
              "-o", os.path.join(d, "pairwise.tsv")])
    pairs = s * (s - 1) // 2
    st.update(junctions=npw, samples=s, pair_columns=pairs, output_bytes=os.path.getsize(os.path.join(d, "pairwise.tsv")),
              p_values_per_s_through_cli=npw * pairs / st["wall_s"],
              values_formatted_per_s=npw * pairs / max(st["writer_format_s"], 1e-9),
              peak_rss_GB=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, rss_before_GB=rss0 / 1e6,
              matrix_in_hbm_GB=npw * pairs * 8 / 1e9)
    doc["pairwise"] = st
    print("pairwise", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
if a.out:
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
