"""`splicedice compare_sample_sets`: two-group rank-sum test per junction of an `_allPS.tsv`.

Drop-in for splicedice/compareSampleSets.py (add_parser :124-159, run_with :163-270): same
flags, same `<3 samples` exit, same output columns
`event mean1 mean2 median1 median2 delta p-value corrected` (+ gene/overlapping/transcript_id
with -a GTF), values printed as numpy float32 / float64 scalars.

On the GPU: the per-row loop :216-232 (NaN drop, <3 skip, scipy ranksums, medians, means)
-> sdice_ranksum; multipletests(..., "fdr_bh") :235 -> sdice_bh over the tested rows.
The GTF annotation columns are host string work, as in the reference.
"""
import sys

import numpy as np

from .engine import Context


def samples_from_manifest(path):
    """First whitespace-separated token of every line (compareSampleSets.py:105-115)."""
    with open(path) as fin:
        return [line.split()[0] for line in fin]


def column_indices(group, cols):
    """Columns of the table that belong to the group, in TABLE order (compareSampleSets.py:96-102)."""
    return np.nonzero(np.isin(cols, group))[0].astype(np.int32)


def read_ps_table(path, as_table=False):
    """`_allPS.tsv` -> (row names array, column names array, float32 matrix), :193-204.  as_table: the row names as a
    textio.NameTable (one byte string + offsets: a million-row table costs no Python string per row)."""
    from . import textio
    header, rows, matrix = textio.read_table_numeric(path, np.float32, as_table=as_table)     # text -> float64 -> float32, as numpy
    headers = header.strip().split("\t")[1:]
    return (rows if as_table else np.array(rows)), np.array(headers), matrix


def read_annotation(gtf_path):
    """GTF -> (junction -> gene names, (chrom,strand) -> {(start,stop): gene names},
    junction -> transcript ids); restates getAnnotated (compareSampleSets.py:32-93)."""
    def attr(info, key):
        return [x[1] for x in info if key in x[0]][0]

    gene_coords, genes, transcripts = {}, {}, {}
    with open(gtf_path) as gtf:
        for line in gtf:
            if line.startswith("#"):
                continue
            row = line.rstrip().split("\t")
            info = [x.split('"') for x in row[8].split(";")]
            chrom, strand = row[0], row[6]
            start, stop = int(row[3]), int(row[4]) - 1
            if row[2] == "transcript":
                tid = attr(info, "transcript_id")
                genes[tid] = attr(info, "gene_name")
                transcripts[(tid, chrom, strand)] = []
            elif row[2] == "exon":
                transcripts[(attr(info, "transcript_id"), chrom, strand)].append((start, stop))
            elif row[2] == "gene":
                gene_name = attr(info, "gene_name")
                attr(info, "gene_id")      # the reference requires the attribute to exist
                gene_coords.setdefault((chrom, strand), {}).setdefault((start, stop), []).append(gene_name)
    annotated, transcript_ids = {}, {}
    for (tid, chromosome, strand), exons in transcripts.items():
        for i in range(len(exons) - 1):
            junction = (chromosome, exons[i][1], exons[i + 1][0], strand)
            if junction in annotated:
                if genes[tid] not in annotated[junction]:
                    annotated[junction].append(genes[tid])
                    transcript_ids[junction].append(tid)
            else:
                annotated[junction] = [genes[tid]]
                transcript_ids[junction] = [tid]
    return annotated, gene_coords, transcript_ids


def annotation_suffixes(names, gtf_path):
    """'\\tgene\\toverlapping\\ttranscript_id' for every event name (compareSampleSets.py:238-264).  The reference
    walks all gene intervals of the event's (chromosome, strand) per event in Python; here that scan is the
    library's threaded interval join (sdice_interval_overlaps) and the known-junction look-ups stay dict
    look-ups; order of the listed genes = the reference's (dict order of the intervals, file order inside)."""
    from . import textio
    annotated, gene_coords, transcript_ids = read_annotation(gtf_path)
    groups = {key: g for g, key in enumerate(gene_coords)}
    grp_ptr = np.zeros(len(groups) + 1, dtype=np.int64)
    lo, hi, key_names = [], [], []
    for g, intervals in enumerate(gene_coords.values()):
        for (gene_start, gene_stop), gene_names in intervals.items():
            lo.append(gene_start)
            hi.append(gene_stop)
            key_names.append(",".join(gene_names))
        grp_ptr[g + 1] = len(lo)
    ev_group = np.empty(len(names), dtype=np.int32)
    ev_a = np.empty(len(names), dtype=np.int64)
    ev_b = np.empty(len(names), dtype=np.int64)
    junctions = []
    for n, name in enumerate(names):
        chromosome, coords, strand = name.split(":")
        start, stop = (int(x) for x in coords.split("-"))
        start -= 1
        stop += 1
        junctions.append((chromosome, start, stop, strand))
        ev_group[n] = groups.get((chromosome, strand), -1)
        ev_a[n] = start
        ev_b[n] = stop
    ptr, idx = textio.interval_overlaps(ev_group, ev_a, ev_b, grp_ptr, lo, hi)
    ptr = ptr.tolist()
    idx = idx.tolist()
    nan = ["nan"]
    return ["\t" + ",".join(annotated.get(j, nan)) + "\t" + ",".join(key_names[k] for k in idx[ptr[n]:ptr[n + 1]]) +
            "\t" + ",".join(transcript_ids.get(j, nan)) for n, j in enumerate(junctions)]


def compare_dev(matrix, g1_idx, g2_idx, ctx):
    """compare() on the HIP engine stage by stage: table up, rank-sum + BH over the tested rows on resident vectors,
    per-row results down (sdice_ranksum + sdice_bh do the same steps inside two host calls)"""
    from . import _stages
    n = matrix.shape[0]
    f32 = ("med1", "med2", "mean1", "mean2", "delta")
    with _stages.stage("h2d"):
        d_ps = ctx.to_device(matrix, np.float32)
        d_g1, d_g2 = ctx.to_device(g1_idx, np.int32), ctx.to_device(g2_idx, np.int32)
        out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64),
                   **{k: ctx.empty(n, np.float32) for k in f32})
        d_q = ctx.empty(n, np.float64)
    with _stages.stage("kernels"):
        ctx.ranksum_dev(d_ps, d_g1, d_g2, out)
        ctx.bh_masked_dev(out["p"], out["tested"], d_q)
        ctx.sync()
    with _stages.stage("d2h"):
        res = {k: v.to_host() for k, v in out.items()}
        q = d_q.to_host()
    for a in (d_ps, d_g1, d_g2, d_q, *out.values()):
        a.free()
    keep = np.flatnonzero(res["tested"])
    r = {k: res[k][keep] for k in ("p",) + f32}
    r["corrected"] = q[keep]
    return keep, r


def compare(matrix, g1_idx, g2_idx, ctx):
    """-> (kept row indices, dict of compacted per-row results incl. BH-corrected p)."""
    if hasattr(ctx, "ranksum_dev") and matrix.shape[0]:
        return compare_dev(matrix, g1_idx, g2_idx, ctx)
    res = ctx.ranksum(matrix, g1_idx, g2_idx)
    keep = np.flatnonzero(res["tested"])
    out = {k: res[k][keep] for k in ("p", "med1", "med2", "mean1", "mean2", "delta")}
    out["corrected"] = ctx.bh(out["p"]) if keep.size else np.zeros(0)
    return keep, out


def compare_sharded(matrix, g1_idx, g2_idx, ctx, L):
    """compare() with the table rows cut into one block per rank (rows are independent here): every rank tests its
    block, the per-row statistics cross the ranks as ONE packed block in ONE all-gather (distributed.stat_layout, padded to
    the longest block), rank 0 corrects."""
    from . import distributed
    n = matrix.shape[0]
    lo, hi = L.row_block(n)
    blocks = [L.row_block(n, r) for r in range(L.world)]
    maxk = max(max(b - a for a, b in blocks), 1)
    res = ctx.ranksum(np.ascontiguousarray(matrix[lo:hi]), g1_idx, g2_idx)
    stats = {k: np.zeros(maxk, dt) for k, dt in zip(distributed.STAT_NAMES, distributed._STAT_DTYPES)}
    for k in distributed.STAT_NAMES:
        stats[k][: hi - lo] = res[k]
    gathered = L.comm(ctx).allgather(distributed.pack_stats_host(stats, maxk))
    host = distributed.unpack_stats_host(gathered, maxk, L.world)
    full = {k: np.concatenate([host[k][r * maxk: r * maxk + b - a] for r, (a, b) in enumerate(blocks)])
            for k in ("tested", "p", "med1", "med2", "mean1", "mean2", "delta")}
    keep = np.flatnonzero(full["tested"])
    out = {k: full[k][keep] for k in ("p", "med1", "med2", "mean1", "mean2", "delta")}
    out["corrected"] = ctx.bh(out["p"]) if (keep.size and L.root) else np.zeros(keep.size)
    return keep, out


def add_parser(parser):
    parser.add_argument("--psiSPLICEDICE", type=str, required=True,
                        help="Compressed NPZ formatted PSI matrix from 'splicedice quant'.")
    parser.add_argument("-m1", "--manifest1", type=str, required=True,
                        help="Manifest containing samples for sample set group1")
    parser.add_argument("-m2", "--manifest2", type=str, required=True,
                        help="Manifest containing samples for sample set group2")
    parser.add_argument("-a", "--annotation", type=str, required=False, default="",
                        help="Optional GTF file to label known splice junctions and genes")
    parser.add_argument("-o", "--outputFile", type=str, required=True,
                        help="Output filename for tab-separated table")


def run_with(args, ctx=None):
    from . import mgpu
    L = mgpu.launcher()             # (reads the torchrun environment before any GPU call)
    g1 = samples_from_manifest(args.manifest1)
    g2 = samples_from_manifest(args.manifest2)
    if len(g1) < 3 or len(g2) < 3:
        print("Cannot conduct wilcoxon with less than 3 samples in either group. Exit.", file=sys.stderr)
        sys.exit(1)

    from . import _stages
    with _stages.stage("parse"):
        rows, cols, matrix = read_ps_table(args.psiSPLICEDICE, as_table=True)
    g1_idx = column_indices(g1, cols)
    g2_idx = column_indices(g2, cols)

    own_ctx = ctx is None
    ctx = ctx if ctx is not None else Context(L.local_rank)
    try:
        keep, r = compare_sharded(matrix, g1_idx, g2_idx, ctx, L) if L.world > 1 else compare(matrix, g1_idx, g2_idx, ctx)
    finally:
        if own_ctx:
            ctx.close()
    if not L.root:
        return                       # one set of output files: rank 0 writes the table

    base_header = "event\tmean1\tmean2\tmedian1\tmedian2\tdelta\tp-value\tcorrected"
    if not args.annotation:
        from . import textio
        # numeric table only: the library's multithreaded formatter (numpy str() of float32 /
        # float64 per column, byte-identical to the reference's print(*fields, sep="\t"))
        with _stages.stage("format+write"):
            textio.write_columns(args.outputFile, base_header + "\n", rows.take(keep),
                                 [r["mean1"], r["mean2"], r["med1"], r["med2"], r["delta"], r["p"], r["corrected"]],
                                 ["repr"] * 7)
        return
    from . import textio
    names = list(rows.take(keep))
    textio.write_columns(args.outputFile, base_header + "\tgene\toverlapping\ttranscript_id\n", names,
                         [r["mean1"], r["mean2"], r["med1"], r["med2"], r["delta"], r["p"], r["corrected"]],
                         ["repr"] * 7, suffixes=annotation_suffixes(names, args.annotation))


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    add_parser(p)
    run_with(p.parse_args())
