"""`splicedice similarity` on the MI355X engine.

Scores every sample of an `_allPS.tsv` table against the significant events of a
compare_sample_sets output (reference: splicedice/similarity.py).  Host side mirrors the
reference's reading rules; the per-sample counting over the PS matrix runs in the HIP kernel
`similarity_kernel` (sdice_similarity).

  * comparison table (similarity.py:5-21): header skipped; a row is used when float(p-value) is
    not > 0.05 and delta != 0; midpoint = median1 - delta/2 (Python floats of the text fields).
  * allps table (:24-47): rows whose name is not in the comparison are ignored; "nan" fields do
    not count; delta < 0 scores ps < midpoint, delta > 0 scores ps > midpoint.
  * output (:57-66): lines sorted by (score, sample, count) descending;
    sample[<TAB>group1<TAB>group2]<TAB>{score/count:0.03f}<TAB>score<TAB>count; with a manifest
    only samples named in it are written.  A sample with count 0 raises ZeroDivisionError as
    the reference does.
"""
import numpy as np

from . import textio
from .engine import Context


def read_vs_file(vs_filename):
    """similarity.py:5-21 -> (midpoints dict, deltas dict)"""
    midpoints, deltas = {}, {}
    with open(vs_filename) as vs_file:
        vs_file.readline()
        for line in vs_file:
            row = line.rstrip().split("\t")
            if float(row[6]) > 0.05:
                continue
            delta = float(row[5])
            if delta == 0:
                continue
            midpoints[row[0]] = float(row[3]) - (delta / 2)
            deltas[row[0]] = delta
    return midpoints, deltas


def row_parameters(names, midpoints, deltas):
    """Per table row: midpoint (float64) and sign of delta (int8; 0 = row not scored)."""
    mid = np.zeros(len(names), dtype=np.float64)
    sign = np.zeros(len(names), dtype=np.int8)
    for i, name in enumerate(names):
        d = deltas.get(name)
        if d is not None:
            mid[i] = midpoints[name]
            sign[i] = -1 if d < 0 else 1
    return mid, sign


def score_samples(ctx, allps_filename, midpoints, deltas):
    """similarity.py:24-47 -> (samples, scores, counts)"""
    header, names, ps = textio.read_table_numeric(allps_filename, dtype=np.float64)
    samples = header.rstrip().split("\t")[1:]
    mid, sign = row_parameters(names, midpoints, deltas)
    scores, counts = ctx.similarity(ps, mid, sign)
    return samples, [int(x) for x in scores], [int(x) for x in counts]


def get_groups(manifest_filename):
    groups = {}
    with open(manifest_filename) as manifest:
        for line in manifest:
            name, path, group1, group2 = line.rstrip().split("\t")
            groups[name] = (group1, group2)
    return groups


def write_scores(output_filename, scores, samples, counts, groups=None):
    with open(output_filename, "w") as score_file:
        for score, sample, count in sorted(zip(scores, samples, counts), reverse=True):
            if groups and sample in groups:
                group1, group2 = groups[sample]
                score_file.write(f"{sample}\t{group1}\t{group2}\t{score / count:0.03f}\t{score}\t{count}\n")
            elif not groups:
                score_file.write(f"{sample}\t{score / count:0.03f}\t{score}\t{count}\n")


def add_parser(parser):
    parser.add_argument("--manifest", "-m", action="store", default=None,
                        help="tab-separated list of samples for group names")
    parser.add_argument("--comparison", "-c", action="store", required=True,
                        help="Output table from compare_sample_sets")
    parser.add_argument("--allps", "-a", action="store", required=True, help="Allps table from splicedice quant")
    parser.add_argument("--output", "-o", action="store", required=True, help="Output filename")


def run_with(args, ctx=None):
    ctx = ctx if ctx is not None else Context(0)
    midpoints, deltas = read_vs_file(args.comparison)
    samples, scores, counts = score_samples(ctx, args.allps, midpoints, deltas)
    groups = get_groups(args.manifest) if args.manifest else None
    write_scores(args.output, scores, samples, counts, groups)
