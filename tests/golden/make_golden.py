#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports the reference's Python modules from /root/reference
(read-only) and records their outputs on small seeded inputs.  The fixtures (inputs +
expected outputs) are data and are committed; the reference itself never travels.

statsmodels is not installed in this image: `multipletests(p, method="fdr_bh")` is
shimmed with scipy.stats.false_discovery_control; outputs that
pass through that call are therefore "parity unpinned" for the BH step only.

    python tests/golden/make_golden.py
"""
import argparse
import contextlib
import io
import json
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from splicedice_amd import synth  # noqa: E402


def install_statsmodels_shim():
    import scipy.stats as st

    def multipletests(pvals, alpha=0.05, method="fdr_bh"):
        assert method == "fdr_bh"
        p = np.asarray(pvals, dtype=float)
        q = st.false_discovery_control(p, method="bh") if p.size else p.copy()
        return (q <= alpha, q, None, None)

    sm = types.ModuleType("statsmodels")
    sms = types.ModuleType("statsmodels.stats")
    smm = types.ModuleType("statsmodels.stats.multitest")
    smm.multipletests = multipletests
    sm.stats = sms
    sms.multitest = smm
    sys.modules["statsmodels"] = sm
    sys.modules["statsmodels.stats"] = sms
    sys.modules["statsmodels.stats.multitest"] = smm


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = fn(*a, **k)
    return r, buf.getvalue()


def ns(**kw):
    return argparse.Namespace(**kw)


def fresh(d):
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    return d


def main():
    install_statsmodels_shim()
    import splicedice.SPLICEDICE as SD
    import splicedice.counts_to_ps as C2P
    import splicedice.compareSampleSets as CSS
    import splicedice.pairwise_fisher as PF
    from scipy.stats import fisher_exact, ranksums

    # ------------------------------------------------------------------ quant (config 1)
    qdir = fresh(os.path.join(HERE, "quant_c1"))
    indir = fresh(os.path.join(qdir, "inputs"))
    files = synth.write_c1_dataset(indir, seed=1, n_target=1000, n_samples=4)
    # a plain bed sample as well (type "bed": name field not e:..;o:..)
    with open(os.path.join(indir, "s4.plain.bed"), "w") as fh, open(files[0][1]) as src:
        for i, line in enumerate(src):
            row = line.rstrip("\n").split("\t")
            if i % 3 == 0:
                fh.write("\t".join([row[0], row[1], row[2], f"j{i}", row[4], row[5]]) + "\n")
    files.append(("s4", os.path.join(indir, "s4.plain.bed")))
    with open(os.path.join(qdir, "manifest.rel.tsv"), "w") as fh:
        for i, (name, path) in enumerate(files):
            fh.write(f"{name}\t{os.path.basename(path)}\tmeta{i}\t{'A' if i % 2 == 0 else 'B'}\n")
    manifest_abs = os.path.join(qdir, "_manifest_abs.tmp")
    with open(manifest_abs, "w") as fh:
        for i, (name, path) in enumerate(files):
            fh.write(f"{name}\t{path}\tmeta{i}\t{'A' if i % 2 == 0 else 'B'}\n")

    variants = {
        "default": dict(),
        "lowcov_drim": dict(lowCoverageNan=True, drim=True, minUnique=8, noMultimap=True),
        "strict": dict(minEntropy=1.2, minOverhang=10, maxLength=15000, minLength=500),
    }
    banners = {}
    for vname, over in variants.items():
        out = fresh(os.path.join(qdir, f"expected_{vname}"))
        args = ns(manifest=manifest_abs, output_prefix=os.path.join(out, "out"),
                  maxLength=50000, minLength=50, minOverhang=5, drim=False, noMultimap=False,
                  filter="gtag_only", minUnique=5, lowCoverageNan=False, minEntropy=1)
        for k, v in over.items():
            setattr(args, k, v)
        SD.Sample.sampleList = []
        SD.Sample.groups = {}
        _, text = quiet(SD.run_with, args)
        banners[vname] = [ln.split("[")[0].rstrip() for ln in text.splitlines()]
        with open(os.path.join(out, "args.json"), "w") as fh:
            json.dump(over, fh)
    with open(os.path.join(qdir, "banners.json"), "w") as fh:
        json.dump(banners, fh, indent=1)
    os.remove(manifest_abs)

    # ------------------------------------------------------------------ quant, hand-made edge cases
    # duplicate lines (last count wins), both strands on the same coordinates, strand '.', annotated
    # junctions that bypass the filters, every filter failing once, SJ.out.tab strand 0 / non-canonical
    # motifs / multimapped counts, chromosome names whose string order differs from numeric order, file
    # types that contribute nothing (.bam, unknown) and a leafcutter file
    edir = fresh(os.path.join(HERE, "quant_edge"))
    ein = fresh(os.path.join(edir, "inputs"))
    sb = "e:1.50:1.20;o:20;m:GT_AG"
    edge_files = {
        "a.junc.bed": [
            f"chr2\t100\t300\t{sb};a:?\t12\t+", f"chr2\t100\t300\t{sb};a:?\t3\t+",
            f"chr10\t150\t400\t{sb};a:?\t9\t-", f"chr10\t150\t400\t{sb};a:?\t9\t+",
            f"chrX\t50\t60000\t{sb};a:?\t20\t+", "chr2\t250\t500\te:0.50:1.20;o:20;m:GT_AG;a:?\t30\t+",
            "chr2\t260\t520\te:1.50:1.20;o:3;m:GT_AG;a:?\t30\t+", "chr2\t270\t530\te:0.10:0.10;o:1;m:GT_AG;a:GENE1\t2\t+",
            f"chr2\t280\t540\t{sb};a:?\t7\t.", f"chrM\t10\t200\t{sb};a:?\t8\t-", f"chr2\t290\t300\t{sb};a:?\t40\t+",
            f"chr1\t5\t120\t{sb};a:?\t5\t+", f"chr1\t60\t200\t{sb};a:?\t6\t+", f"chr1\t120\t180\t{sb};a:?\t7\t+",
            f"chr1\t200\t260\t{sb};a:?\t8\t+"],
        "b.SJ.out.tab": [
            "chr2\t101\t300\t1\t1\t0\t4\t3\t30", "chr2\t271\t530\t1\t1\t1\t10\t0\t20", "chr10\t151\t400\t2\t2\t0\t6\t0\t25",
            "chr10\t151\t400\t0\t1\t0\t50\t0\t25", "chr2\t301\t900\t1\t0\t0\t50\t0\t25", "chr2\t301\t900\t1\t3\t0\t50\t0\t25",
            "chrX\t1001\t2000\t2\t2\t1\t0\t9\t12", "chr1\t61\t200\t1\t1\t0\t2\t1\t12", "chr1\t6\t120\t1\t1\t0\t50\t50\t40"],
        "c.plain.bed": [
            "chr2\t100\t300\tj1\t6\t+", "chr2\t100\t300\tj1\t0\t+", "chr10\t150\t400\tj2\t5\t-", "chr3\t10\t20\tj3\t100\t+",
            "chr2\t400\t700\tj4\t4\t+", "chr2\t400\t700\tj5\t8\t-", "chr1\t120\t180\tj6\t11\t+"],
        "d.bam": [],
        "e.leafcutter.junc": ["chr2\t400\t700\tclu_1\t9\t+", "chrX\t1000\t2000\tclu_2\t5\t-", "chr1\t200\t260\tclu_3\t1\t+"],
        "f.counts.txt": ["whatever\t1\t2"],
    }
    for fname, lines in edge_files.items():
        with open(os.path.join(ein, fname), "w") as fh:
            fh.write("".join(line + "\n" for line in lines))
    with open(os.path.join(edir, "manifest.rel.tsv"), "w") as fh:
        for i, fname in enumerate(edge_files):
            fh.write(f"s{i}\t{fname}\tm{i}\t{'A' if i < 3 else 'B'}\n")
    manifest_abs = os.path.join(edir, "_manifest_abs.tmp")
    with open(manifest_abs, "w") as fh:
        for i, fname in enumerate(edge_files):
            fh.write(f"s{i}\t{os.path.join(ein, fname)}\tm{i}\t{'A' if i < 3 else 'B'}\n")
    edge_variants = {
        "default": dict(),
        "nomulti_lowcov": dict(noMultimap=True, lowCoverageNan=True, minUnique=6, drim=True),
        "short": dict(minLength=5, maxLength=100000, minOverhang=1, minEntropy=0.05, minUnique=1),
    }
    for vname, over in edge_variants.items():
        out = fresh(os.path.join(edir, f"expected_{vname}"))
        args = ns(manifest=manifest_abs, output_prefix=os.path.join(out, "out"),
                  maxLength=50000, minLength=50, minOverhang=5, drim=False, noMultimap=False,
                  filter="gtag_only", minUnique=5, lowCoverageNan=False, minEntropy=1)
        for k, v in over.items():
            setattr(args, k, v)
        SD.Sample.sampleList = []
        SD.Sample.groups = {}
        quiet(SD.run_with, args)
        with open(os.path.join(out, "args.json"), "w") as fh:
            json.dump(over, fh)
    os.remove(manifest_abs)

    # ------------------------------------------------------------------ counts_to_ps
    cdir = fresh(os.path.join(HERE, "counts_to_ps"))
    base = os.path.join(qdir, "expected_default", "out")
    for mode in ("c", "r"):
        out = fresh(os.path.join(cdir, f"expected_{mode}"))
        args = ns(clusters=base + "_allClusters.tsv" if mode == "c" else None,
                  recluster=(mode == "r"), inclusion_counts=base + "_inclusionCounts.tsv",
                  output_prefix=os.path.join(out, "out"))
        quiet(C2P.run_with, args)

    # ------------------------------------------------------------------ direct kernels: clusters + psi arrays
    adir = fresh(os.path.join(HERE, "arrays"))
    for tag, n, s, seed, kw in (("a", 400, 7, 11, {}),
                                ("b", 1500, 5, 12, dict(gene_spacing=6000, len_span=30000)),
                                ("c", 64, 3, 13, dict(n_chrom=2, gene_spacing=100000)),
                                ("dense", 300, 4, 14, dict(n_chrom=1, gene_spacing=100, len_span=3000))):
        cr, left, right, strand = synth.make_junctions(n, seed, **({"n_chrom": 5} | kw))
        names = sorted(f"chr{i + 1}" for i in range(int(cr.max()) + 1))  # rank -> name (string order)
        tuples = [(names[cr[i]], int(left[i]), int(right[i]), synth.STRANDS[strand[i]]) for i in range(n)]
        obj = SD.SPLICEDICE.__new__(SD.SPLICEDICE)
        obj.junctions = set(tuples)
        obj.clusters = obj.getClusters()
        obj.junctionIndex = {j: i for i, j in enumerate(sorted(obj.clusters))}
        counts = synth.make_counts(n, s, seed + 100)
        row_of = np.array([obj.junctionIndex[t] for t in tuples], dtype=np.int32)
        counts_rows = np.zeros_like(counts)
        counts_rows[row_of] = counts          # counts in output row order
        obj.counts = counts_rows.astype(np.float32)
        obj.manifest = [None] * s
        obj.args = ns(lowCoverageNan=False)
        obj.low = []
        with np.errstate(all="ignore"):
            psi = obj.calculatePsi()
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        col = []
        for r, j in enumerate(sorted(obj.clusters)):
            lst = obj.clusters[j]
            row_ptr[r + 1] = row_ptr[r] + len(lst)
            col.extend(obj.junctionIndex[o] for o in lst)
        np.savez_compressed(os.path.join(adir, f"cluster_psi_{tag}.npz"),
                            chrom_rank=cr, left=left, right=right, strand=strand,
                            counts_rows=counts_rows, row_of=row_of, row_ptr=row_ptr,
                            col=np.asarray(col, dtype=np.int32), psi=psi)

    # ------------------------------------------------------------------ compare_sample_sets
    mdir = fresh(os.path.join(HERE, "compare"))
    rng = np.random.default_rng(21)
    n, s = 300, 12
    ps = synth.make_ps_matrix(n, s, seed=21, shift_frac=0.3, nan_frac=0.12)
    ps[5, :] = np.float32(0.5)                 # all ties
    ps[6, :6] = np.nan                         # group 1 all NaN -> skipped
    ps[7, :4] = np.nan                         # only 2 valid in g1 -> skipped
    ps[8, :3] = np.nan                         # exactly 3 valid in g1 -> tested
    ps[9, :] = np.float32([0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1])
    ps[10, :] = np.float32([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 0.95, 0.85])
    samples = [f"samp{i}" for i in range(s)]
    with open(os.path.join(mdir, "in_allPS.tsv"), "w") as fh:
        fh.write("cluster\t" + "\t".join(samples) + "\n")
        for r in range(n):
            fh.write(f"chr1:{1000 + 10 * r}-{2000 + 10 * r}:+\t" + "\t".join(f"{x:.3f}" for x in ps[r]) + "\n")
    with open(os.path.join(mdir, "m1.tsv"), "w") as fh:
        for i in (0, 1, 2, 3, 4, 5):
            fh.write(f"{samples[i]}\tpath\tmeta\tA\n")
        fh.write("not_in_table\tpath\tmeta\tA\n")
    with open(os.path.join(mdir, "m2.tsv"), "w") as fh:
        for i in (11, 6, 7, 8, 9, 10):        # order in manifest must not matter
            fh.write(f"{samples[i]} path meta B\n")
    quiet(CSS.run_with, ns(psiSPLICEDICE=os.path.join(mdir, "in_allPS.tsv"),
                           manifest1=os.path.join(mdir, "m1.tsv"), manifest2=os.path.join(mdir, "m2.tsv"),
                           annotation="", outputFile=os.path.join(mdir, "expected_out.tsv")))
    # tiny GTF for the annotation branch
    with open(os.path.join(mdir, "anno.gtf"), "w") as fh:
        fh.write("# test\n")
        fh.write('chr1\tt\tgene\t900\t2500\t.\t+\t.\tgene_id "G1"; gene_name "GENEA";\n')
        fh.write('chr1\tt\ttranscript\t900\t2500\t.\t+\t.\tgene_id "G1"; transcript_id "T1"; gene_name "GENEA";\n')
        fh.write('chr1\tt\texon\t900\t1000\t.\t+\t.\tgene_id "G1"; transcript_id "T1"; gene_name "GENEA";\n')
        fh.write('chr1\tt\texon\t2002\t2500\t.\t+\t.\tgene_id "G1"; transcript_id "T1"; gene_name "GENEA";\n')
        fh.write('chr1\tt\tgene\t3000\t4200\t.\t+\t.\tgene_id "G2"; gene_name "GENEB";\n')
    quiet(CSS.run_with, ns(psiSPLICEDICE=os.path.join(mdir, "in_allPS.tsv"),
                           manifest1=os.path.join(mdir, "m1.tsv"), manifest2=os.path.join(mdir, "m2.tsv"),
                           annotation=os.path.join(mdir, "anno.gtf"),
                           outputFile=os.path.join(mdir, "expected_out_gtf.tsv")))

    # ------------------------------------------------------------------ pairwise
    pdir = fresh(os.path.join(HERE, "pairwise"))
    cr, left, right, strand = synth.make_junctions(48, 31, n_chrom=2)
    names = ["chr1", "chr2"]
    tuples = sorted((names[cr[i]], int(left[i]), int(right[i]), synth.STRANDS[strand[i]]) for i in range(48))
    obj = SD.SPLICEDICE.__new__(SD.SPLICEDICE)
    obj.junctions = set(tuples)
    clusters = obj.getClusters()
    counts = synth.make_counts(48, 6, 32, mean=20)
    counts[3, :] = 0
    jstr = lambda j: f"{j[0]}:{j[1]}-{j[2]}:{j[3]}"
    with open(os.path.join(pdir, "in_inclusionCounts.tsv"), "w") as fh:
        fh.write("cluster\t" + "\t".join(f"p{i}" for i in range(6)) + "\n")
        for r, j in enumerate(tuples):
            fh.write(jstr(j) + "\t" + "\t".join(f"{x:.0f}" for x in counts[r]) + "\n")
    with open(os.path.join(pdir, "in_allClusters.tsv"), "w") as fh:
        for j in tuples:
            print(jstr(j) + "\t" + ",".join(jstr(o) for o in clusters[j]), file=fh)
    with open(os.path.join(pdir, "filter.txt"), "w") as fh:
        for j in tuples[::2]:
            fh.write(jstr(j) + "\n")
    for mode in ("pairwise", "all", "none"):
        quiet(PF.run_with, ns(inclusionSPLICEDICE=os.path.join(pdir, "in_inclusionCounts.tsv"),
                              clusters=os.path.join(pdir, "in_allClusters.tsv"), chi2=False,
                              multiple_test_correction=mode, filter_list=None,
                              output=os.path.join(pdir, f"expected_{mode}.tsv")))
    quiet(PF.run_with, ns(inclusionSPLICEDICE=os.path.join(pdir, "in_inclusionCounts.tsv"),
                          clusters=os.path.join(pdir, "in_allClusters.tsv"), chi2=False,
                          multiple_test_correction="none", filter_list=os.path.join(pdir, "filter.txt"),
                          output=os.path.join(pdir, "expected_none_filtered.tsv")))

    # --chi2 needs every expected frequency > 0 (scipy raises otherwise and the reference run aborts):
    # keep the junctions that have overlaps and make every count positive
    kept = [r for r, j in enumerate(tuples) if clusters[j]]
    with open(os.path.join(pdir, "in_inclusionCounts_pos.tsv"), "w") as fh:
        fh.write("cluster\t" + "\t".join(f"p{i}" for i in range(6)) + "\n")
        for r in kept:
            fh.write(jstr(tuples[r]) + "\t" + "\t".join(f"{x + 1:.0f}" for x in counts[r]) + "\n")
    for mode in ("none", "pairwise"):
        quiet(PF.run_with, ns(inclusionSPLICEDICE=os.path.join(pdir, "in_inclusionCounts_pos.tsv"),
                              clusters=os.path.join(pdir, "in_allClusters.tsv"), chi2=True,
                              multiple_test_correction=mode, filter_list=None,
                              output=os.path.join(pdir, f"expected_chi2_{mode}.tsv")))
    try:
        quiet(PF.run_with, ns(inclusionSPLICEDICE=os.path.join(pdir, "in_inclusionCounts.tsv"),
                              clusters=os.path.join(pdir, "in_allClusters.tsv"), chi2=True,
                              multiple_test_correction="none", filter_list=None,
                              output=os.path.join(pdir, "_should_not_exist.tsv")))
        outcome = "completed"
    except ValueError as e:
        outcome = "ValueError: " + str(e)[:80]
    with open(os.path.join(pdir, "chi2_on_zero_rows.json"), "w") as fh:
        json.dump({"outcome": outcome}, fh)
    if os.path.exists(os.path.join(pdir, "_should_not_exist.tsv")):
        os.remove(os.path.join(pdir, "_should_not_exist.tsv"))

    # ------------------------------------------------------------------ KATs (scipy call sites)
    kats = []
    edge = [(0, 0, 5, 7), (0, 5, 0, 7), (3, 4, 0, 0), (5, 0, 7, 0), (1, 1, 1, 1), (10, 10, 10, 10),
            (7, 3, 3, 7), (3, 7, 7, 3), (12, 5, 5, 12), (0, 10, 10, 0), (10, 0, 0, 10), (30, 60, 210, 190),
            (1, 0, 0, 1), (2, 0, 0, 0), (100, 1, 1, 100), (500, 20, 30, 600), (1000, 1200, 1100, 1300),
            (10000, 12000, 11000, 9000), (20000, 20000, 20000, 20000), (5, 1000, 1000, 5),
            (0, 30, 200, 180), (30, 0, 200, 180), (17, 23, 0, 180), (16777215, 3, 5, 16777215),
            (2000, 1, 1, 2000), (150, 150, 150, 150), (6, 2, 1, 4), (1, 9, 11, 3), (3, 1, 9, 11)]
    rng = np.random.default_rng(41)
    for _ in range(400):
        scale = int(rng.choice([5, 40, 300, 3000]))
        edge.append(tuple(int(x) for x in rng.integers(0, scale, size=4)))
    for _ in range(120):                       # symmetric-margin tables: exact pmf ties
        a, b = (int(x) for x in rng.integers(0, 60, size=2))
        edge.append((a, b, b, a))
        m = int(rng.integers(1, 40))
        edge.append((a, m, a, m + int(rng.integers(0, 3))))
    for t in edge:
        p = float(fisher_exact([[t[0], t[1]], [t[2], t[3]]])[1])
        kats.append([list(t), p])
    with open(os.path.join(HERE, "kat_fisher.json"), "w") as fh:
        json.dump(kats, fh)

    rk = []
    cases = [(np.float32([0.5] * 3), np.float32([0.5] * 3)),
             (np.float32([0.1, 0.2, 0.3]), np.float32([0.4, 0.5, 0.6])),
             (np.float32([0, 0, 1, 1]), np.float32([0, 1, 1, 1, 0.5]))]
    for n1, n2 in ((3, 3), (5, 50), (50, 50), (500, 500), (37, 64), (128, 3)):
        x = np.round(rng.random(n1), 3).astype(np.float32)
        y = np.round(np.clip(rng.random(n2) + 0.1, 0, 1), 3).astype(np.float32)
        cases.append((x, y))
    for x, y in cases:
        z, p = ranksums(x, y)
        rk.append(dict(x=[float(v) for v in x], y=[float(v) for v in y], z=float(z), p=float(p),
                       med1=float(np.median(x)), med2=float(np.median(y)),
                       mean1=float(np.mean(x)), mean2=float(np.mean(y))))
    with open(os.path.join(HERE, "kat_ranksums.json"), "w") as fh:
        json.dump(rk, fh)

    # str() formats of numpy scalars as the reference prints them (compareSampleSets.py:270)
    vals32 = np.float32([0.17800002, 0.5, 1.0, 0.0, 1e-5, 0.33333334, 123456.79, 2.5e-8, 0.1, 0.30000001])
    vals64 = np.float64([0.376759117811582, 1.0, 0.0, 4.9817526009363926e-11, 1e-300, 0.1, 1e16, 123456.789, 5e-324])
    with open(os.path.join(HERE, "kat_numpy_str.json"), "w") as fh:
        json.dump(dict(f32=[[float(v), str(v)] for v in vals32],
                       f64=[[float(v).hex(), str(v)] for v in vals64]), fh)
    # ------------------------------------------------------------------ similarity
    import splicedice.similarity as SIM
    sdir = fresh(os.path.join(HERE, "similarity"))
    cdir = os.path.join(HERE, "compare")
    # (a) the compare fixture scored against its own PS table, without and with a group manifest
    quiet(SIM.run_with, ns(comparison=os.path.join(cdir, "expected_out.tsv"), allps=os.path.join(cdir, "in_allPS.tsv"),
                           manifest=None, output=os.path.join(sdir, "expected_scores.tsv")))
    with open(os.path.join(sdir, "groups.tsv"), "w") as fh:
        for i in range(0, 12, 2):
            fh.write(f"samp{i}\tpath{i}\tg{i % 3}\tbatch{i % 4}\n")
    quiet(SIM.run_with, ns(comparison=os.path.join(cdir, "expected_out.tsv"), allps=os.path.join(cdir, "in_allPS.tsv"),
                           manifest=os.path.join(sdir, "groups.tsv"), output=os.path.join(sdir, "expected_scores_groups.tsv")))
    # (b) hand-made comparison table: midpoints that coincide with 3-decimal PS values (strict
    # comparisons in float64 of the TEXT values), p above the cut-off, zero delta, negative delta
    with open(os.path.join(sdir, "in_vs.tsv"), "w") as fh:
        fh.write("event\tmean1\tmean2\tmedian1\tmedian2\tdelta\tp-value\tcorrected\n")
        fh.write("e0\t0.5\t0.4\t0.5\t0.4\t0.1\t0.01\t0.02\n")            # midpoint 0.45, delta > 0
        fh.write("e1\t0.2\t0.6\t0.2\t0.6\t-0.4\t0.001\t0.002\n")         # midpoint 0.4, delta < 0
        fh.write("e2\t0.2\t0.6\t0.2\t0.6\t-0.4\t0.06\t0.5\n")            # p > 0.05 -> ignored
        fh.write("e3\t0.3\t0.3\t0.3\t0.3\t0.0\t0.01\t0.02\n")            # delta == 0 -> ignored
        fh.write("e4\t0.17800002\t0.5\t0.167\t0.5545\t-0.3875\t0.05\t0.05\n")  # p == 0.05 kept
        fh.write("not_in_table\t0.1\t0.2\t0.1\t0.2\t-0.1\t0.01\t0.01\n")
    with open(os.path.join(sdir, "in_allPS.tsv"), "w") as fh:
        fh.write("cluster\ta\tb\tc\td\te\n")
        fh.write("e0\t0.450\t0.449\t0.451\tnan\t1.000\n")
        fh.write("e1\t0.400\t0.399\t0.401\t0.000\tnan\n")
        fh.write("e2\t0.100\t0.100\t0.100\t0.100\t0.100\n")
        fh.write("e3\t0.100\t0.100\t0.100\t0.100\t0.100\n")
        fh.write("e4\t0.361\t0.360\t0.36075\tnan\t0.362\n")
        fh.write("extra\t0.5\t0.5\t0.5\t0.5\t0.5\n")
    quiet(SIM.run_with, ns(comparison=os.path.join(sdir, "in_vs.tsv"), allps=os.path.join(sdir, "in_allPS.tsv"),
                           manifest=None, output=os.path.join(sdir, "expected_scores_handmade.tsv")))
    # ------------------------------------------------------------------ findOutliers
    import splicedice.findOutliers as FO
    odir = fresh(os.path.join(HERE, "outliers"))
    rng = np.random.default_rng(55)
    n_ev, n_s = 400, 16
    cols = np.array([f"samp{i}" for i in range(n_s)])
    rows = np.array([f"chr2:{1000 + 13 * r}-{5000 + 17 * r}:-" for r in range(n_ev)])
    base = rng.beta(0.7, 0.7, size=(n_ev, 1))
    mat = np.clip(base + rng.normal(0, 0.03, size=(n_ev, n_s)), 0, 1)
    mat = np.round(mat, 3)
    spikes = rng.integers(0, n_ev, size=60)
    mat[spikes, rng.integers(0, n_s, size=60)] = np.round(rng.random(60), 3)      # outlying samples
    mat[rng.random((n_ev, n_s)) < 0.04] = np.nan
    mat[7, :12] = np.nan                      # > 20 % NaN among the null group -> skipped
    mat[8, :] = 0.5                           # std 0 -> skipped
    mat[9, :] = np.nan
    with open(os.path.join(odir, "samples.tsv"), "w") as fh:
        for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
            fh.write(f"samp{i}\tpath{i}\n")
        fh.write("not_in_table\tp\n")
    with open(os.path.join(odir, "null.tsv"), "w") as fh:
        for i in (15, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):          # order must not matter
            fh.write(f"samp{i}\tpath{i}\n")
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        np.savez(os.path.join(odir, f"matrix_{tag}.npz"), cols=cols, rows=rows, data=mat.astype(dt))
        for null, name in ((None, "self"), (os.path.join(odir, "null.tsv"), "null")):
            _, text = quiet(FO.run_with, ns(psiSPLICEDICE=os.path.join(odir, f"matrix_{tag}.npz"),
                                            manifest=os.path.join(odir, "samples.tsv"), nullMan=null, outlierCutoff=3,
                                            dpsiThrsh=0.1))
            with open(os.path.join(odir, f"expected_{tag}_{name}.txt"), "w") as fh:
                fh.write(text)
    _, text = quiet(FO.run_with, ns(psiSPLICEDICE=os.path.join(odir, "matrix_f32.npz"),
                                    manifest=os.path.join(odir, "samples.tsv"), nullMan=None, outlierCutoff=2, dpsiThrsh=0.1))
    with open(os.path.join(odir, "expected_f32_cutoff2.txt"), "w") as fh:
        fh.write(text)
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
