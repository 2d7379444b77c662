#!/usr/bin/env python3
"""Time the REAL reference (imported from /root/reference, build container only, 1 thread as it
ships) and the oracle's restatement on the same seeded inputs -- BASELINE.md section 3,
"reference, in-container".  Shows that the port timed beside the GPU numbers (bench.py
cpu_baseline, kind "port") runs at the reference's own speed.

    python tests/golden/time_reference.py  ->  profiles/reference_in_container.json
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_statsmodels_shim, quiet, REPO  # noqa: E402  (also puts /root/reference on sys.path)

from oracle import oracle_np as O  # noqa: E402
from splicedice_amd import synth  # noqa: E402


def main():
    install_statsmodels_shim()
    import splicedice.SPLICEDICE as SD
    from scipy.stats import fisher_exact, ranksums
    out = {"host": f"{os.cpu_count()} vCPU build container, 1 thread used", "numpy": np.__version__}

    # ---- C2 sub-sample: clustering + PS, 100 k junctions x 100 samples
    n, s = 100_000, 100
    cr, left, right, strand = synth.make_junctions(n, 7)
    names = synth.chrom_names(24)
    counts = synth.make_counts(n, s, 20)
    obj = SD.SPLICEDICE.__new__(SD.SPLICEDICE)
    obj.junctions = set((names[cr[i]], int(left[i]), int(right[i]), synth.STRANDS[strand[i]]) for i in range(n))
    t = time.time()
    obj.clusters = obj.getClusters()
    t_cluster = time.time() - t
    order = sorted(obj.clusters)
    obj.junctionIndex = {j: i for i, j in enumerate(order)}
    obj.counts = counts.astype(np.float32)
    obj.manifest = list(range(s))
    obj.args = type("A", (), {"lowCoverageNan": False})()
    obj.low = set()
    t = time.time()
    quiet(obj.calculatePsi)
    t_psi = time.time() - t
    t = time.time()
    row_of, row_ptr, col = O.cluster_csr(cr, left, right, strand)
    t_ocl = time.time() - t
    t = time.time()
    O.calculate_psi(counts, row_ptr, col)
    t_opsi = time.time() - t
    out["quant_100k_x_100"] = {
        "reference": {"getClusters_s": round(t_cluster, 2), "calculatePsi_s": round(t_psi, 2),
                      "entries_per_s": n * s / (t_cluster + t_psi)},
        "oracle_port": {"cluster_csr_s": round(t_ocl, 2), "calculate_psi_s": round(t_opsi, 2),
                        "entries_per_s": n * s / (t_ocl + t_opsi)}}

    # ---- C3 sub-sample: rank-sum loop, 20 k rows, 50 v 50
    m = 20_000
    ps = synth.make_ps_matrix(m, 100, 3)
    g1, g2 = np.arange(0, 50), np.arange(50, 100)
    t = time.time()
    for r in range(m):                       # compareSampleSets.py:216-232
        d1, d2 = ps[r, g1], ps[r, g2]
        d1, d2 = d1[~np.isnan(d1)], d2[~np.isnan(d2)]
        if len(d1) < 3 or len(d2) < 3:
            continue
        ranksums(d1, d2)
        np.median(d1); np.median(d2); np.mean(d1); np.mean(d2)
    t_ref = time.time() - t
    t = time.time()
    O.compare_rows(ps, g1.astype(np.int32), g2.astype(np.int32))
    t_or = time.time() - t
    out["compare_20k_rows_50v50"] = {"reference_loop_rows_per_s": m / t_ref, "oracle_port_rows_per_s": m / t_or}

    # ---- C4 sub-sample: Fisher loop, 20 junctions x 60 samples (all pairs)
    nj, sc = 20, 60
    incl = synth.make_counts(nj, sc, 40)
    excl = (synth.make_counts(nj, sc, 41).astype(np.int64)) * 6
    t = time.time()
    for r in range(nj):                      # pairwise_fisher.py:164-179
        for i in range(sc - 1):
            for j in range(i + 1, sc):
                fisher_exact([[incl[r, i], excl[r, i]], [incl[r, j], excl[r, j]]])
    t_ref = time.time() - t
    t = time.time()
    O.fisher_pairs(incl, excl)
    t_or = time.time() - t
    nt = nj * sc * (sc - 1) // 2
    out["pairwise_fisher_%d_tables" % nt] = {"reference_loop_p_per_s": nt / t_ref, "oracle_port_p_per_s": nt / t_or}

    path = os.path.join(REPO, "profiles", "reference_in_container.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
