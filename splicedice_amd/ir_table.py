"""`splicedice ir_table`: intron-retention table from intron coverage + inclusion counts.

Drop-in for the reference sub-command (splicedice/ir_table.py: add_parser :7-35, run_with :165-197): same
flags, stdout lines and output files `<prefix>_intron_retention.tsv` (+ `_intron_retention_RSD.tsv` with
-r).  For every junction of a sample's `<sample>_intron_coverage.txt` (format of intron_coverage.py:221-230)

    IR = median / (median + count[junction] + sum of the counts of its mutually exclusive junctions)

(ir_table.py:119-133).  The cluster sum is the same primitive as the exclusion sum of PS and pairwise: it
runs on the GPU once for the whole count table (sdice_ps with the `excl` output over the CSR of the
cluster file), the per-line arithmetic and the text stay on the host as in the reference.

Behaviour kept, including the reference's quirks: samples come in os.listdir order; a neighbour missing
from the count table is reported as "mxCluster ..." and skipped; a junction missing from a sample's counts,
a junction without a line in the cluster file and a sample without a count column each print "cluster ..."
and end that sample's file (the rows it then lacks make the writers die on a KeyError, as in the reference);
count tables that are not integer-valued are summed in float64 in the reference's order; 0/0 is NaN; junctions are kept when ANY sample's RSD is
below the threshold, which needs -r (without it the reference dies on RSD[sample][junction]: KeyError, so
does this).
"""
import os

import numpy as np

from . import textio
from .engine import Context


def add_parser(parser):
    """Flag surface of ir_table.py:7-35."""
    parser.add_argument("-i", "--inclusionCounts", action="store", help="")
    parser.add_argument("-c", "--clusters", action="store",
                        help="allClusters.tsv file with mutually exclusive clusters for each junction.")
    parser.add_argument("-d", "--coverageDirectory", action="store", help="")
    parser.add_argument("-o", "--outputPrefix", action="store", help="")
    parser.add_argument("-r", "--makeRSDtable", action="store_true",
                        help="Make a table of relative standard deviations in coverage across intron.")
    parser.add_argument("-s", "--singleJunctionCalculation", action="store_true",
                        help="Calculate IR value using individual junction counts, and not count of all junctions in cluster.")
    parser.add_argument("-a", "--annotation", action="store", help="GTF file with gene annotation.")
    parser.add_argument("-j", "--allJunctions", action="store_true",
                        help="Output IR values for all junctions above RSD threshold. Default: only annotated junctions")
    parser.add_argument("-t", "--RSDthreshold", default=1.0, action="store", help="RSD cutoff for inclusion. Default: 1.0.")


def get_annotated(annotation):
    """names 'chrom:exon_end-next_exon_start-1:strand' of every annotated intron (ir_table.py:38-68)"""
    genes, transcripts = {}, {}
    with open(annotation) as gtf:
        for line in gtf:
            if line.startswith("#"):
                continue
            row = line.rstrip().split("\t")
            if row[2] == "transcript":
                info = [x.split('"') for x in row[8].split(";")]
                tid = [x[1] for x in info if "transcript_id" in x[0]][0]
                try:
                    gid = [x[1] for x in info if "gene_name" in x[0]][0]
                except IndexError:
                    gid = [x[1] for x in info if "gene_id" in x[0]][0]
                genes[tid] = gid
                transcripts[(tid, row[0], row[6])] = []
            if row[2] == "exon":
                info = [x.split('"') for x in row[8].split(";")]
                tid = [x[1] for x in info if "transcript_id" in x[0]][0]
                transcripts[(tid, row[0], row[6])].append((int(row[3]), int(row[4])))
    annotated = set()
    for (tid, chromosome, strand), exons in transcripts.items():
        for i in range(len(exons) - 1):
            annotated.add(f"{chromosome}:{exons[i][1]}-{exons[i + 1][0] - 1}:{strand}")
    return annotated


def get_clusters(filename):
    """junction -> list of mutually exclusive junction names (ir_table.py:83-92)"""
    clusters = {}
    with open(filename) as fh:
        for line in fh:
            row = line.strip().split("\t")
            try:
                clusters[row[0]] = row[1].split(",")
            except IndexError:
                clusters[row[0]] = []
    return clusters


def cluster_sums(ctx, counts, index, clusters, wanted, own_first=False):
    """sum of the count rows of the neighbours of event r that are present in the table (every listed name counts, as
    the reference's loop adds them one by one) -- on the GPU for the whole table at once.  Only the events in `wanted`
    get a list (the reference never looks the others up); an event without a line in the cluster file gets none (the
    caller reports it as the reference does).  Integer tables: int64 sums of the neighbours.  Float tables
    (`own_first`): float64 sums in the reference's order -- the event's own count first, then the neighbours in list
    order, one IEEE addition each (ir_table.py:122-126)."""
    n = counts.shape[0]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    col = []
    lists = {}
    for name in wanted:
        r = index.get(name)
        if r is None or name not in clusters:
            continue
        lists[r] = ([r] if own_first else []) + [index[mx] for mx in clusters[name] if mx in index]
    for r in range(n):
        col.extend(lists.get(r, ()))
        row_ptr[r + 1] = len(col)
    col = np.asarray(col, dtype=np.int32)
    if own_first:
        return ctx.excl_f64(counts, row_ptr, col)
    return ctx.ps(counts, row_ptr, col, want_excl=True, want_ps=False)


def run_with(args, ctx=None):
    import time
    start = time.time()
    samples = [s.replace("_intron_coverage.txt", "") for s in os.listdir(args.coverageDirectory)
               if s.endswith("intron_coverage.txt")]
    print("Gathering inclusion counts and clusters...")
    header, events, mat = textio.read_table_numeric(args.inclusionCounts, np.float64)
    table_samples = header.rstrip().split("\t")[1:]
    # integer-valued tables take the integer kernel; anything else the reference reads with float() (normalised or
    # fractional counts) goes through the float64 sums in the reference's order of additions
    mat = np.asarray(mat, dtype=np.float64)
    integral = bool(mat.size == 0 or (np.isfinite(mat).all() and (mat >= 0).all() and (mat < 2 ** 31).all()
                                      and (mat == np.floor(mat)).all()))
    counts = mat.astype(np.int32) if integral else mat
    index = {name: r for r, name in enumerate(events)}          # (a repeated row name: the last one wins, as the dict does)
    column = {s: c for c, s in enumerate(table_samples)}
    annotated = get_annotated(args.annotation) if not args.allJunctions else None
    clusters = get_clusters(args.clusters) if not args.singleJunctionCalculation else None

    print("Calculating IR values...")
    # pass 1 over the coverage files: the lines the reference looks at
    lines = {}
    wanted = set()
    for sample in samples:
        rec = []
        with open(os.path.join(args.coverageDirectory, f"{sample}_intron_coverage.txt")) as fh:
            for line in fh:
                row = line.strip().split("\t")
                cluster = f"{row[0]}:{row[1]}-{row[2]}:{row[5]}"
                if not args.allJunctions and cluster not in annotated:
                    continue
                rec.append((cluster, float(row[4]), row[-1]))
                wanted.add(cluster)
        lines[sample] = rec
    excl = None
    if clusters is not None and events:
        own_ctx = ctx is None
        ctx = ctx if ctx is not None else Context(0)
        try:
            excl = cluster_sums(ctx, counts, index, clusters, wanted, own_first=not integral)
        finally:
            if own_ctx:
                ctx.close()
    # pass 2: the reference's per-line arithmetic and messages, in its order (ir_table.py:96-138)
    IR, RSD, junctions = {}, {}, set()
    for sample in samples:
        IR[sample], RSD[sample] = {}, {}
        col_s = column.get(sample)
        for cluster, median, cov_text in lines[sample]:
            junctions.add(cluster)
            if args.makeRSDtable:
                cov = np.array(cov_text.split(",")).astype(float)
                with np.errstate(all="ignore"):
                    RSD[sample][cluster] = np.std(cov) / np.mean(cov)
            r = index.get(cluster)
            # the reference's outer try (ir_table.py:121-136): a sample without a count column, a junction without a count
            # row, a junction without a line in the cluster file -- each a KeyError there -- print this and end the sample
            if r is None or col_s is None or (clusters is not None and cluster not in clusters):
                print("cluster", sample, cluster)
                break
            intron = float(counts[r, col_s])
            if clusters is not None:
                for mx in clusters[cluster]:
                    if mx not in index:
                        print("mxCluster", sample, cluster, mx)
                intron = float(excl[r, col_s]) if not integral else intron + float(excl[r, col_s])
            try:
                IR[sample][cluster] = median / (median + intron)
            except ZeroDivisionError:
                IR[sample][cluster] = np.nan
    filtered = []
    for junction in junctions:
        for sample in samples:
            if RSD[sample][junction] < args.RSDthreshold:      # KeyError without -r, as the reference
                filtered.append(junction)
                break
    print("Done", time.time() - start)
    print("Writing output...")
    tab = "\t"
    with open(f"{args.outputPrefix}_intron_retention.tsv", "w") as out:
        out.write(f"Junction\t{tab.join(samples)}\n")
        for junction in sorted(filtered):
            out.write(f"{junction}\t{tab.join(f'{IR[sample][junction]:0.03f}' for sample in samples)}\n")
    if args.makeRSDtable:
        with open(f"{args.outputPrefix}_intron_retention_RSD.tsv", "w") as out:
            out.write("Junction\t" + tab.join(f"{sample}_RSD" for sample in samples) + "\n")
            for junction in sorted(filtered):
                out.write(f"{junction}\t{tab.join(f'{RSD[s][junction]:0.03f}' for s in samples)}\n")


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    add_parser(p)
    run_with(p.parse_args())
