#!/usr/bin/env python3
"""counts_to_ps on a FRACTIONAL count table, by RUNNING THE REFERENCE (build container only).

The reference parses `_inclusionCounts.tsv` with dtype=float (counts_to_ps.py:50), so normalised counts
are legal input; its sums then round and the order of the additions (list order, :63-67) shows in the
last bit.  The table is quant_c1's count table scaled per sample by an awkward factor and printed with
6 decimals, plus a few cells that make 0/0 (nan) and values of very different magnitude in one cluster.

    python tests/golden/make_golden_fractional.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (puts /root/reference on sys.path)


def main():
    import splicedice.counts_to_ps as C2P
    base = os.path.join(HERE, "quant_c1", "expected_default", "out")
    d = MG.fresh(os.path.join(HERE, "counts_to_ps_fractional"))
    rng = np.random.default_rng(77)
    src = open(base + "_inclusionCounts.tsv").read().splitlines()
    header, rows = src[0], [ln.split("\t") for ln in src[1:]]
    s = len(rows[0]) - 1
    factor = rng.uniform(0.013, 3.7, size=s)
    table = os.path.join(d, "in_inclusionCounts.tsv")
    with open(table, "w") as fh:
        fh.write(header + "\n")
        for i, row in enumerate(rows):
            vals = np.array(row[1:], dtype=float) * factor
            if i % 17 == 3:
                vals = vals * 1e9          # one huge member of a cluster: the small ones vanish in the sum
            if i % 23 == 5:
                vals = vals * 1e-7
            cells = [f"{v:.6f}" if i % 5 else f"{v:.3e}" for v in vals]
            fh.write(row[0] + "\t" + "\t".join(cells) + "\n")
    for mode in ("c", "r"):
        out = MG.fresh(os.path.join(d, f"expected_{mode}"))
        args = MG.ns(clusters=base + "_allClusters.tsv" if mode == "c" else None, recluster=(mode == "r"),
                     inclusion_counts=table, output_prefix=os.path.join(out, "out"))
        MG.quiet(C2P.run_with, args)
    # (the -c run read quant_c1's cluster file; the -r run's cluster file equals counts_to_ps/expected_r's)
    os.remove(os.path.join(d, "expected_r", "out_allClusters.tsv"))
    print(open(os.path.join(d, "expected_c", "out_allPS.tsv")).read()[:400])


if __name__ == "__main__":
    main()
