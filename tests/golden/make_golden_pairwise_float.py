#!/usr/bin/env python3
"""`pairwise` on a FRACTIONAL count table, by RUNNING THE REFERENCE (build container only).

The reference parses the count table with dtype=float (pairwise_fisher.py:46-61), adds the rows of an event's
cluster as floats in table order (:158-160) and hands float 2x2 tables to scipy.stats.fisher_exact, which casts
them to int64 (truncation).  The table is tests/golden/pairwise's count table scaled per sample by an awkward
factor, plus planted cells whose float sums land on / just below an integer (0.1 + 0.2 + 0.7, 0.3 + 0.6 + 0.1, ...).

    python tests/golden/make_golden_pairwise_float.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (puts /root/reference on sys.path, installs nothing by itself)


def main():
    MG.install_statsmodels_shim()
    import splicedice.pairwise_fisher as PF
    src_dir = os.path.join(HERE, "pairwise")
    d = MG.fresh(os.path.join(HERE, "pairwise_fractional"))
    rng = np.random.default_rng(123)
    src = open(os.path.join(src_dir, "in_inclusionCounts.tsv")).read().splitlines()
    header, rows = src[0], [ln.split("\t") for ln in src[1:]]
    s = len(rows[0]) - 1
    factor = rng.uniform(0.21, 2.9, size=s)
    table = os.path.join(d, "in_inclusionCounts.tsv")
    planted = [0.1, 0.2, 0.7, 0.3, 0.6, 0.1, 1.1, 2.2, 3.3, 0.7, 0.2, 0.1]
    with open(table, "w") as fh:
        fh.write(header + "\n")
        for i, row in enumerate(rows):
            vals = np.array(row[1:], dtype=float) * factor
            if i % 4 == 1:                      # sums that sit on the edge of an integer, order dependent
                vals[0] = planted[i % len(planted)]
                vals[1] = planted[(i + 5) % len(planted)] + 3
            fh.write(row[0] + "\t" + "\t".join(repr(float(v)) for v in vals) + "\n")
    with open(os.path.join(src_dir, "in_allClusters.tsv")) as a, open(os.path.join(d, "in_allClusters.tsv"), "w") as b:
        b.write(a.read())
    for mode in ("none", "pairwise"):
        MG.quiet(PF.run_with, MG.ns(inclusionSPLICEDICE=table, clusters=os.path.join(d, "in_allClusters.tsv"), chi2=False,
                                    multiple_test_correction=mode, filter_list=None,
                                    output=os.path.join(d, f"expected_{mode}.tsv")))
    print(open(os.path.join(d, "expected_none.tsv")).read()[:300])


if __name__ == "__main__":
    main()
