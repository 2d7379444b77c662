"""`splicedice similarity` on the MI355X engine.

Scores every sample of an `_allPS.tsv` table against the significant events of a
compare_sample_sets output.  The per-sample counting over the PS matrix runs in the HIP kernel
`similarity_kernel` (sdice_similarity); this module is the host side and keeps the reference's
reading and writing rules (splicedice/similarity.py, line numbers below):

  comparison table (:5-21)  first line skipped; an event is used unless float(p-value) > 0.05 or
                            delta == 0; its midpoint is median1 - delta/2 on Python floats of the text
  PS table (:24-47)         events absent from the comparison are ignored, "nan" fields do not count;
                            delta < 0 scores ps < midpoint, delta > 0 scores ps > midpoint
  output (:57-66)           samples ordered by (score, name, count) descending; one line
                            name[<TAB>group1<TAB>group2]<TAB>score/count as 0.03f<TAB>score<TAB>count;
                            with a manifest only the samples it names; count == 0 raises
                            ZeroDivisionError exactly as the reference does
"""
import numpy as np

from . import textio
from .engine import Context

P_CUTOFF = 0.05
COL_MEDIAN1, COL_DELTA, COL_P = 3, 5, 6


def significant_events(comparison_path):
    """-> {event: (midpoint, delta)} for the rows of a compare_sample_sets table that pass the cut-off"""
    events = {}
    with open(comparison_path) as table:
        next(table, None)
        for record in table:
            cells = record.rstrip().split("\t")
            delta = float(cells[COL_DELTA])
            if float(cells[COL_P]) > P_CUTOFF or delta == 0:
                continue
            events[cells[0]] = (float(cells[COL_MEDIAN1]) - delta / 2, delta)
    return events


def row_parameters(names, events):
    """Per PS-table row: midpoint (float64) and sign of delta (int8; 0 = row is not scored)."""
    mid = np.zeros(len(names), dtype=np.float64)
    sign = np.zeros(len(names), dtype=np.int8)
    for i, name in enumerate(names):
        hit = events.get(name)
        if hit is not None:
            mid[i] = hit[0]
            sign[i] = 1 if hit[1] > 0 else -1
    return mid, sign


def score_table(ctx, allps_path, events):
    """-> (sample names, scores, counts) -- the PS matrix is parsed as float64 by the library's reader
    and scored on the GPU"""
    header, names, ps = textio.read_table_numeric(allps_path, dtype=np.float64)
    mid, sign = row_parameters(names, events)
    scores, counts = ctx.similarity(ps, mid, sign)
    return header.rstrip().split("\t")[1:], scores.tolist(), counts.tolist()


def sample_groups(manifest_path):
    with open(manifest_path) as manifest:
        rows = (line.rstrip().split("\t") for line in manifest)
        return {name: (g1, g2) for name, _path, g1, g2 in rows}


def write_report(path, samples, scores, counts, groups=None):
    ranking = sorted(range(len(samples)), key=lambda i: (scores[i], samples[i], counts[i]), reverse=True)
    with open(path, "w") as out:
        for i in ranking:
            name, score, count = samples[i], scores[i], counts[i]
            if groups:
                if name not in groups:
                    continue
                label = "\t".join((name,) + tuple(groups[name]))
            else:
                label = name
            out.write(f"{label}\t{score / count:0.03f}\t{score}\t{count}\n")


def add_parser(parser):
    parser.add_argument("--manifest", "-m", action="store", default=None,
                        help="tab-separated list of samples for group names")
    parser.add_argument("--comparison", "-c", action="store", required=True,
                        help="Output table from compare_sample_sets")
    parser.add_argument("--allps", "-a", action="store", required=True, help="Allps table from splicedice quant")
    parser.add_argument("--output", "-o", action="store", required=True, help="Output filename")


def run_with(args, ctx=None):
    ctx = ctx if ctx is not None else Context(0)
    events = significant_events(args.comparison)
    samples, scores, counts = score_table(ctx, args.allps, events)
    write_report(args.output, samples, scores, counts, sample_groups(args.manifest) if args.manifest else None)
