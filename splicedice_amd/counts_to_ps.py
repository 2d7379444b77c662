"""`splicedice counts_to_ps`: PS table from an `_inclusionCounts.tsv` and clusters.

Drop-in for splicedice/counts_to_ps.py (add_parser :81-94, run_with :96-113): `-c` reads an
`_allClusters.tsv`, `-r` re-derives the clusters from the junction names of the count table
and writes `<prefix>_allClusters.tsv` (sorted by raw string, :76); both write
`<prefix>_allPS.tsv` in tuple order (:61).

On the GPU: determine_clusters (:16-41) -> sdice_cluster; the per-junction sum of the
overlapping count rows and the quotient (:63-68) -> sdice_ps_f64: float64 throughout, sums in
list order, as the reference's `dtype=float` rows are -- fractional / normalised count tables
are legal input here, and this sub-command prints the float64 quotient itself rather than a
float32 as quant does.
"""
import numpy as np

from . import textio
from .engine import Context


def get_clusters(cluster_file):
    """name -> list of overlap names ('' for an empty list), counts_to_ps.py:8-14."""
    clusters = {}
    with open(cluster_file) as cf:
        for line in cf:
            junction, overlaps = line.rstrip("\n").split("\t")
            clusters[junction] = overlaps.split(",")
    return clusters


def determine_clusters(counts_file, ctx):
    """Clusters of the junctions named in the count table (counts_to_ps.py:16-41), on the GPU."""
    junctions = []
    with open(counts_file) as cfile:
        cfile.readline()
        for line in cfile:
            junctions.append(textio.parse_junction_name(line.split("\t", 1)[0]))
    unique = list(dict.fromkeys(junctions))        # the reference's dict keys collapse duplicates
    _, cr, left, right, strand = textio.junction_arrays(unique)
    row_of, row_ptr, col = ctx.cluster(cr, left, right, strand)
    names = [None] * len(unique)
    for i, r in enumerate(row_of):
        names[r] = textio.junction_name(unique[i])
    return {names[r]: [names[c] for c in col[row_ptr[r]:row_ptr[r + 1]]] for r in range(len(names))}


def get_counts(count_file):
    """header line + name -> row index, float64 matrix (counts_to_ps.py:43-51: dtype=float)."""
    header, names, data = textio.read_table_numeric(count_file, np.float64)
    index = {}
    for i, name in enumerate(names):
        index[name] = i                           # a repeated name keeps its last row, as a dict would
    return header, index, np.ascontiguousarray(data, dtype=np.float64)


def write_ps_values(clusters, header, index, counts, output_prefix, ctx):
    """counts_to_ps.py:58-70: ps = own / (own + sum of overlap rows), '0.3f', tuple order."""
    j_list = sorted(clusters.keys(), key=textio.parse_junction_name)
    row_of_name = {name: r for r, name in enumerate(j_list)}
    rows = np.fromiter((index[name] for name in j_list), dtype=np.int64, count=len(j_list))   # KeyError as the reference
    row_ptr = np.zeros(len(j_list) + 1, dtype=np.int64)
    col = []
    extra = []          # overlap names that are in the count table but are not cluster keys
    extra_index = {}
    for r, name in enumerate(j_list):
        for overlap in clusters[name]:
            if overlap == "":
                continue
            c = row_of_name.get(overlap)
            if c is None:
                if overlap not in extra_index:
                    extra_index[overlap] = len(j_list) + len(extra)
                    extra.append(index[overlap])   # KeyError as the reference
                c = extra_index[overlap]
            col.append(c)
        row_ptr[r + 1] = len(col)
    all_rows = np.concatenate([rows, np.asarray(extra, dtype=np.int64)]) if extra else rows
    table = counts[all_rows] if len(all_rows) else np.zeros((0, counts.shape[1] if counts.ndim == 2 else 0), np.float64)
    ps = ctx.ps_f64(table, row_ptr, np.asarray(col, dtype=np.int32), n_out=len(j_list))
    textio.write_table(f"{output_prefix}_allPS.tsv", header, j_list, ps, ".3f")      # f"{x:0.3f}" on float64


def write_clusters(clusters, output_prefix):
    with open(f"{output_prefix}_allClusters.tsv", "w") as cluster_file:
        for junction in sorted(clusters):
            cluster_file.write(f"{junction}\t{','.join(clusters[junction])}\n")


def add_parser(parser):
    parser.add_argument("--clusters", "-c", action="store", default=None, help="allClusters.tsv file from SPLICEDICE")
    parser.add_argument("--recluster", "-r", action="store_true",
                        help="Determine clusters from splice junctions in counts file")
    parser.add_argument("--inclusion_counts", "-i", action="store", required=True,
                        help="inclusionCounts.tsv file from SPLICEDICE")
    parser.add_argument("--output_prefix", "-o", action="store", required=True,
                        help="output filename path and prefix")


def run_with(args, ctx=None):
    own_ctx = ctx is None
    ctx = ctx if ctx is not None else Context(0)
    try:
        if args.clusters:
            print("Gathering clusters...")
            clusters = get_clusters(args.clusters)
        elif args.recluster:
            print("Determining clusters from counts file...")
            clusters = determine_clusters(args.inclusion_counts, ctx)
            write_clusters(clusters, args.output_prefix)
        else:
            # the reference falls through to a NameError at counts_to_ps.py:112
            raise SystemExit("counts_to_ps: one of --clusters/-c or --recluster/-r is required")
        print("Gathering counts...")
        header, index, counts = get_counts(args.inclusion_counts)
        print("Calculating PS values...")
        write_ps_values(clusters, header, index, counts, args.output_prefix, ctx)
        print("Done.")
    finally:
        if own_ctx:
            ctx.close()


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser(description="Calculate PS values with inclusion count file.")
    add_parser(p)
    run_with(p.parse_args())
