"""`splicedice findOutliers` on the MI355X engine.

Reports (event, sample) pairs whose PS lies far from a background ("null") group of samples.  The
row statistics -- np.nanmean and np.nanstd of every event over the null columns, in the matrix's own
dtype -- come from the HIP kernel `rowstats_kernel` (sdice_rowstats), bit-identical to numpy; the
thresholds and the report are applied here exactly as the reference does (findOutliers.py:98-152):

  * sample names are the first tab-separated field of every manifest line (not stripped);
    the null group defaults to the same manifest; fewer than 10 null samples -> message on stderr, exit 1
  * input is an NPZ file with `cols` (sample names), `rows` (event names), `data` (matrix)
  * an event is skipped when more than 20 % of its null values are NaN or when its null std < 0.001
  * z = (value - mean) / std in the matrix dtype; a line is printed when z >= --outlierCutoff and
    |z - mean| >= --outlierCutoff  (the reference compares the z-score, not delta-PS, with the
    second threshold and never reads --dpsiThrsh; kept as is)
  * line: event, sample, z, value, mean, std separated by tabs, numpy scalar formatting
"""
import sys

import numpy as np

from .engine import Context


def add_parser(parser):
    parser.add_argument("--psiSPLICEDICE", type=str, action="store", required=True,
                        help="Compressed NPZ formatted PSI matrix from quantSPLICEDICE.")
    parser.add_argument("-m", "--manifest", action="store", required=True, help="Sample manifest for samples you want ")
    parser.add_argument("--nullMan", action="store", required=False, default=None,
                        help="List of samples you want to use to use as baseline/background. Samples must be in table.")
    parser.add_argument("--outlierCutoff", type=int, action="store", required=False, default=3,
                        help="Report events where zScore >= N (default 3).")
    parser.add_argument("--dpsiThrsh", type=float, action="store", required=False, default=0.1,
                        help="Report events where dPSI >= N (default 0.1).")


def first_fields(path):
    with open(path) as fh:
        return [line.split("\t")[0] for line in fh]


def load_matrix(path):
    try:
        return np.load(path)
    except Exception:
        print("ERR ** Cannot load matrix %s. Check path or format." % path)
        sys.exit(1)


def find_hits(ctx, matrix, null_idx, sample_idx, cutoff):
    """-> (event positions, positions inside sample_idx, z, mean, std) of the reported pairs, row-major"""
    if matrix.dtype not in (np.float32, np.float64):
        matrix = matrix.astype(np.float64)
    empty = np.zeros(0, dtype=np.int64)
    if null_idx.size == 0 or sample_idx.size == 0 or matrix.shape[0] == 0:
        return empty, empty, matrix[:0, :0], matrix[:0, 0], matrix[:0, 0]
    mean, std, n_nan = ctx.rowstats(matrix, null_idx)
    with np.errstate(invalid="ignore", divide="ignore"):
        usable = ~(n_nan / null_idx.size > 0.2) & ~(std < 0.001)
        vals = matrix[:, sample_idx]
        z = (vals - mean[:, None]) / std[:, None]
        hit = (z >= cutoff) & (np.abs(z - mean[:, None]) >= cutoff) & usable[:, None]
    ev, pos = np.nonzero(hit)
    return ev, pos, z, mean, std


def run_with(args, ctx=None):
    samples = first_fields(args.manifest)
    null = samples if args.nullMan is None else first_fields(args.nullMan)
    if len(null) < 10:
        print("Less than 10 samples detected in null group. Too few. Exit.", file=sys.stderr)
        sys.exit(1)
    data = load_matrix(args.psiSPLICEDICE)
    cols, rows, matrix = data["cols"], data["rows"], data["data"]
    null_idx = np.flatnonzero(np.isin(cols, null))
    sample_idx = np.flatnonzero(np.isin(cols, samples))
    ctx = ctx if ctx is not None else Context(0)
    if matrix.dtype not in (np.float32, np.float64):
        matrix = matrix.astype(np.float64)
    ev, pos, z, mean, std = find_hits(ctx, matrix, null_idx, sample_idx, args.outlierCutoff)
    for e, p in zip(ev.tolist(), pos.tolist()):
        print(rows[e], cols[sample_idx[p]], z[e, p], matrix[e, sample_idx[p]], mean[e], std[e], sep="\t")
