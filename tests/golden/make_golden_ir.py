#!/usr/bin/env python3
"""Golden fixtures for `splicedice ir_table` by RUNNING THE REFERENCE module.

Build-container only.  The reference's ir_table.py uses `np.float` (ir_table.py:118), an alias that numpy
removed in 1.24; the installed numpy is 2.2, so the alias is restored (`np.float = float`, exactly what
numpy < 1.24 defined) before the module runs -- the reference source is not touched.  Inputs: the
reference-generated quant fixtures (tests/golden/quant_c1/expected_default) + synthetic
`<sample>_intron_coverage.txt` files in the format intron_coverage.py:221-230 writes + a small GTF.

    python tests/golden/make_golden_ir.py
"""
import argparse
import contextlib
import io
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def main():
    np.float = float                                    # numpy < 1.24 alias used by ir_table.py:118
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ir_table", os.path.join(REF, "splicedice", "ir_table.py"))
    IT = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(IT)

    out = os.path.join(HERE, "ir_table")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "coverage"))
    q = os.path.join(HERE, "quant_c1", "expected_default")
    counts_path, clusters_path = os.path.join(out, "in_inclusionCounts.tsv"), os.path.join(out, "in_allClusters.tsv")
    with open(os.path.join(q, "out_inclusionCounts.tsv")) as fh:
        lines = fh.read().splitlines()
    header, rows = lines[0], lines[1:]
    samples = header.split("\t")[1:]
    all_names = [r.split("\t")[0] for r in rows]
    rng = np.random.default_rng(11)
    chosen = [all_names[i] for i in sorted(rng.choice(len(all_names), size=90, replace=False))]
    # one NEIGHBOUR of a chosen junction is dropped from the count table: the reference then prints
    # "mxCluster <sample> <junction> <neighbour>" for it and goes on without its counts (ir_table.py:125-128)
    neighbours = {}
    with open(os.path.join(q, "out_allClusters.tsv")) as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            neighbours[f[0]] = [x for x in f[1].split(",") if x] if len(f) > 1 else []
    dropped = next(nb for name in chosen for nb in neighbours[name] if nb not in chosen)
    with open(counts_path, "w") as fh:
        fh.write(header + "\n" + "\n".join(r for r in rows if r.split("\t")[0] != dropped) + "\n")
    shutil.copy(os.path.join(q, "out_allClusters.tsv"), clusters_path)
    # coverage files: chrom, left, right, '.', median, strand, percentile positions, counts at them
    for si, s in enumerate(samples):
        with open(os.path.join(out, "coverage", f"{s}_intron_coverage.txt"), "w") as fh:
            for ji, name in enumerate(chosen):
                chrom, coords, strand = name.split(":")
                left, right = coords.split("-")
                cov = rng.poisson(6.0 if ji % 3 else 0.7, size=10)
                if ji == 5:
                    cov[:] = 0                           # all-zero coverage: RSD = nan, and median 0
                median = int(np.median(cov))
                perc = ",".join(str(int(left) + k * (int(right) - int(left)) // 10) for k in range(10))
                fh.write(f"{chrom}\t{left}\t{right}\t.\t{median}\t{strand}\t{perc}\t{','.join(str(int(c)) for c in cov)}\n")
    # GTF: every other chosen junction is an annotated intron (exon end = left, next exon start = right + 1)
    with open(os.path.join(out, "anno.gtf"), "w") as fh:
        fh.write("# synthetic annotation\n")
        for ji, name in enumerate(chosen[::2]):
            chrom, coords, strand = name.split(":")
            left, right = (int(x) for x in coords.split("-"))
            attr = f'gene_id "G{ji}"; transcript_id "T{ji}"; ' + (f'gene_name "N{ji}";' if ji % 2 else "")
            fh.write(f"{chrom}\tsyn\ttranscript\t{left - 200}\t{right + 300}\t.\t{strand}\t.\t{attr}\n")
            fh.write(f"{chrom}\tsyn\texon\t{left - 200}\t{left}\t.\t{strand}\t.\t{attr}\n")
            fh.write(f"{chrom}\tsyn\texon\t{right + 1}\t{right + 300}\t.\t{strand}\t.\t{attr}\n")
    variants = {"annotated": dict(allJunctions=False, singleJunctionCalculation=False),
                "all": dict(allJunctions=True, singleJunctionCalculation=False),
                "all_single": dict(allJunctions=True, singleJunctionCalculation=True)}
    for tag, kw in variants.items():
        prefix = os.path.join(out, f"expected_{tag}")
        args = argparse.Namespace(inclusionCounts=counts_path, clusters=clusters_path, coverageDirectory=os.path.join(out, "coverage"),
                                  outputPrefix=prefix, makeRSDtable=True, annotation=os.path.join(out, "anno.gtf"),
                                  RSDthreshold=1.0, **kw)
        buf = io.StringIO()
        import warnings
        with contextlib.redirect_stdout(buf), np.errstate(all="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            IT.run_with(args)
        lines_out = [ln for ln in buf.getvalue().splitlines() if not ln.startswith("Done")]     # (elapsed time varies)
        with open(prefix + "_stdout.txt", "w") as fh:
            fh.write("\n".join(lines_out) + "\n")
        print(tag, "->", sorted(f for f in os.listdir(out) if f.startswith(f"expected_{tag}")))
    edge_cases(IT, out, counts_path, clusters_path, chosen)


def edge_cases(IT, base, counts_path, clusters_path, chosen):
    """tests/golden/ir_table_edge: what the reference does (stdout, exception, files) when a junction has no line in the
    cluster file, when a coverage file belongs to a sample without a count column, and on a fractional count table"""
    import warnings
    out = os.path.join(HERE, "ir_table_edge")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    cov = os.path.join(base, "coverage")
    # (a) the 7th chosen junction loses its line in the cluster file
    victim = chosen[6]
    with open(clusters_path) as fh, open(os.path.join(out, "noline_allClusters.tsv"), "w") as dst:
        dst.write("".join(line for line in fh if line.split("\t")[0].rstrip("\n") != victim))
    # (b) coverage directory with one more sample than the count table has columns
    ghost = os.path.join(out, "coverage_ghost")
    shutil.copytree(cov, ghost)
    first = sorted(f for f in os.listdir(cov) if f.endswith("_intron_coverage.txt"))[0]
    shutil.copy(os.path.join(cov, first), os.path.join(ghost, "ghost_intron_coverage.txt"))
    # (c) a normalised (fractional) count table
    with open(counts_path) as fh, open(os.path.join(out, "frac_inclusionCounts.tsv"), "w") as dst:
        dst.write(fh.readline())
        for k, line in enumerate(fh):
            f = line.rstrip("\n").split("\t")
            dst.write(f[0] + "\t" + "\t".join(repr(float(v) * (0.37 + 0.01 * (k % 7)) + (0.125 if k % 5 == 0 else 0.0)) for v in f[1:]) + "\n")
    cases = {"noline": dict(inclusionCounts=counts_path, clusters=os.path.join(out, "noline_allClusters.tsv"), coverageDirectory=cov),
             "ghost": dict(inclusionCounts=counts_path, clusters=clusters_path, coverageDirectory=ghost),
             "frac": dict(inclusionCounts=os.path.join(out, "frac_inclusionCounts.tsv"), clusters=clusters_path, coverageDirectory=cov)}
    for tag, kw in cases.items():
        prefix = os.path.join(out, f"expected_{tag}")
        args = argparse.Namespace(outputPrefix=prefix, makeRSDtable=True, annotation=os.path.join(base, "anno.gtf"), RSDthreshold=1.0,
                                  allJunctions=True, singleJunctionCalculation=False, **kw)
        buf, exc = io.StringIO(), "none"
        with contextlib.redirect_stdout(buf), np.errstate(all="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                IT.run_with(args)
            except Exception as e:                      # noqa: BLE001 (the point is to record what the reference raises)
                exc = type(e).__name__
        lines_out = [ln for ln in buf.getvalue().splitlines() if not ln.startswith("Done")]
        with open(prefix + "_stdout.txt", "w") as fh:
            fh.write("\n".join(lines_out) + "\n")
        with open(prefix + "_exception.txt", "w") as fh:
            fh.write(exc + "\n")
        print("edge", tag, exc, "->", sorted(f for f in os.listdir(out) if f.startswith(f"expected_{tag}")))


if __name__ == "__main__":
    main()
