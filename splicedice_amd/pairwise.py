"""`splicedice pairwise`: Fisher's exact test between every pair of samples, per junction.

Drop-in for splicedice/pairwise_fisher.py (add_parser :71-111, run_with :114-200): same flags,
same stdout lines, same output table `clusterID <a_b columns>` with str(float64) cells.

On the GPU: the exclusion gather :158-160 (an O(N) name scan per event in the reference)
-> sdice_ps (integer sums over a CSR built once on the host from the name lists); the per-pair
scipy fisher_exact :164-179 -> sdice_fisher_pairs (or, with --chi2, chi2_contingency :133-136 ->
sdice_chi2_pairs); Benjamini-Hochberg :182-193 -> sdice_bh / sdice_bh_columns.
"""
import numpy as np

from . import textio
from .engine import Context


def get_clusters(filename):
    """event -> list of overlap names; whitespace split, a lone name means no overlaps
    (pairwise_fisher.py:26-43; its filter_list argument never changes the result)."""
    clusters = {}
    with open(filename) as fh:
        for line in fh:
            try:
                event, mxes = line.rstrip().split()
                mxes = mxes.split(",")
            except ValueError:
                event = line.strip()
                mxes = []
            clusters[event] = mxes
    return clusters


def get_event_counts(filename, filter_list=None):
    """samples, events, counts; `-f` keeps only listed rows (pairwise_fisher.py:46-61).  counts: int32 when every cell
    is a non-negative integer (the usual `_inclusionCounts.tsv`), else the float64 table as parsed -- the reference
    reads with dtype=float and fractional / normalised counts are legal there (see fractional_tables)."""
    header, events, mat = textio.read_table_numeric(filename, np.float64)
    samples = header.rstrip().split("\t")[1:]
    if filter_list is not None:
        keep = [i for i, e in enumerate(events) if e in filter_list]
        events = [events[i] for i in keep]
        mat = mat[keep] if keep else np.zeros((0, len(samples)))
    try:
        return samples, events, textio.counts_to_int32(mat, filename)
    except ValueError:
        return samples, events, np.ascontiguousarray(mat, dtype=np.float64)


def fractional_tables(ctx, counts, row_ptr, col):
    """`pairwise` on a table with non-integer cells, as the reference computes it: the exclusion counts are FLOAT sums
    over the event's rows in table order (np.sum(counts[mask], axis=0), pairwise_fisher.py:158-160), and
    scipy.stats.fisher_exact casts the 2x2 table to int64 -- truncation towards zero of the float inclusion count
    and of the float SUM (not the sum of truncated counts).  -> (incl int32[n,s], excl int64[n,s]) for the Fisher kernel."""
    n = counts.shape[0]
    # every event's rows in ascending (table) order: the order of the additions shows in the last bit of a sum, and a
    # sum that lands within an ulp of an integer truncates differently
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(row_ptr))
    order = np.lexsort((col, rows))
    excl_f = ctx.excl_f64(counts, row_ptr, np.ascontiguousarray(col[order], dtype=np.int32))
    incl_t, excl_t = np.trunc(counts), np.trunc(excl_f)
    if not (np.isfinite(incl_t).all() and np.isfinite(excl_t).all()):
        raise ValueError("pairwise: the count table holds NaN or infinite cells")
    if (incl_t < 0).any() or (excl_t < 0).any():
        raise ValueError("All values in `table` must be nonnegative.")        # scipy.stats.fisher_exact's own words
    if (incl_t >= 2.0 ** 31).any() or (excl_t >= 2.0 ** 62).any():
        raise ValueError("pairwise: counts must stay below 2**31")
    return incl_t.astype(np.int32), excl_t.astype(np.int64)


def exclusion_csr(events, clusters):
    """CSR over table rows: for event n, every table row whose name is in clusters[event]
    (np.isin(events, clusters[events[n]]), pairwise_fisher.py:158: each matching row once)."""
    rows_of = {}
    for r, name in enumerate(events):
        rows_of.setdefault(name, []).append(r)
    row_ptr = np.zeros(len(events) + 1, dtype=np.int64)
    col = []
    for n, name in enumerate(events):
        for other in dict.fromkeys(clusters[name]):       # KeyError for an unknown event, as the reference
            col.extend(rows_of.get(other, ()))
        row_ptr[n + 1] = len(col)
    return row_ptr, np.asarray(col, dtype=np.int32)


SLAB_BYTES = 256 << 20      # host memory of the streamed output (rows x pairs x 8 B per slab)


def device_pipeline(ctx, counts, row_ptr, col, chi2, correction, events, header, path, excl=None):
    """exclusion sums -> per-pair test -> correction with the [n, pairs] p-value matrix RESIDENT IN HBM
    (config 4: 200 000 x 19 900 doubles = 32 GB; the reference holds it in host memory,
    pairwise_fisher.py:123-193), then streamed to the output table in row slabs: device -> host ->
    formatter -> file, so host memory stays bounded by SLAB_BYTES whatever the table size."""
    from . import _stages
    n, s = counts.shape
    pairs = s * (s - 1) // 2
    with _stages.stage("h2d"):
        d_counts = ctx.to_device(counts, np.int32)
        d_rp = ctx.to_device(row_ptr, np.int64)
        d_col = ctx.to_device(col if col.size else np.zeros(1, np.int32), np.int32)
    t_kernels = _stages.stage("kernels")
    t_kernels.__enter__()
    if excl is None:
        d_excl = ctx.empty((n, s), np.int64)
        ctx.ps_dev(d_counts, d_rp, d_col, d_excl, None)
    else:
        d_excl = ctx.to_device(excl, np.int64)            # (fractional table: truncated float sums, fractional_tables)
    d_p = ctx.empty((n, pairs), np.float64)
    if chi2:
        d_bad = ctx.empty(1, np.int64)
        ctx.chi2_pairs_dev(d_counts, d_excl, d_p, d_bad)
        n_bad = int(d_bad.to_host()[0])
        if n_bad:
            # scipy.stats.chi2_contingency raises on the first such table and the reference
            # run dies with it (pairwise_fisher.py:167-179)
            raise ValueError("The internally computed table of expected frequencies has a zero element "
                             f"({n_bad} of {n * pairs} sample-pair tables have an empty row or column)")
    else:
        ctx.fisher_pairs_dev(d_counts, d_excl, d_p)
    for a in (d_counts, d_rp, d_col, d_excl):
        a.free()
    if correction == "all":
        d_q = ctx.empty((n, pairs), np.float64)
        ctx.bh_dev(d_p.offset(0, (n * pairs,)), d_q.offset(0, (n * pairs,)))
        ctx.sync()
        d_p.free()
        d_p = d_q
    elif correction == "pairwise":
        ctx.bh_columns_dev(d_p)
    ctx.sync()
    t_kernels.__exit__(None, None, None)
    # two host slabs: the next one comes down (a thread of its own; the library call releases the GIL) while the
    # current one is formatted and written
    from concurrent.futures import ThreadPoolExecutor
    slab = max(1, SLAB_BYTES // (2 * pairs * 8))
    starts = list(range(0, n, slab))
    bufs = [np.empty((min(slab, n), pairs), np.float64) for _ in range(min(2, len(starts)))]

    def fetch(i):
        r0 = starts[i]
        k = min(slab, n - r0)
        return d_p.offset(r0 * pairs, (k, pairs)).to_host(out=bufs[i % 2][:k])

    try:
        with ThreadPoolExecutor(1) as pool:
            nxt = pool.submit(fetch, 0) if starts else None
            for i, r0 in enumerate(starts):
                with _stages.stage("d2h"):
                    host = nxt.result()
                nxt = pool.submit(fetch, i + 1) if i + 1 < len(starts) else None
                with _stages.stage("format+write"):
                    textio.write_table(path, header if r0 == 0 else "", events[r0:r0 + host.shape[0]], host, "repr", append=r0 > 0)
    finally:
        textio.trim()
    d_p.free()


def add_parser(parser):
    # same flags, defaults and choices as pairwise_fisher.py:71-111
    parser.add_argument("--inclusionSPLICEDICE", type=str, required=True,
                        help="inclusion count table written by `splicedice quant` (*_inclusionCounts.tsv)")
    parser.add_argument("-c", "--clusters", type=str, required=True,
                        help="cluster table written by `splicedice quant` (*_allClusters.tsv)")
    parser.add_argument("--chi2", action="store_true", default=False,
                        help="Yates-corrected chi-square per pair instead of Fisher's exact test")
    parser.add_argument("--multiple_test_correction", default="pairwise", choices=["pairwise", "all", "none"],
                        help="Benjamini-Hochberg scope: per sample pair (default), over all p-values, or off")
    parser.add_argument("-f", "--filter_list", help="text file with one event per line: only these rows are analysed")
    parser.add_argument("-o", "--output", default="pairwise.tsv", help="output table (tab separated)")


def run_with(args, ctx=None):
    from . import mgpu
    L = mgpu.launcher()             # (reads the torchrun environment before any GPU call)
    if args.filter_list is not None:
        with open(args.filter_list, "r") as fh:
            filter_list = set(line.rstrip() for line in fh)
    else:
        filter_list = None

    from . import _stages
    with _stages.stage("parse"):
        samples, events, counts = get_event_counts(args.inclusionSPLICEDICE, filter_list)
    print("Counts loaded from", args.inclusionSPLICEDICE, "...")
    with _stages.stage("parse"):
        clusters = get_clusters(args.clusters)
    print("Clusters loaded from", args.clusters, "...")
    pairs = [(i, j) for i in range(len(samples) - 1) for j in range(i + 1, len(samples))]
    columns = [f"{samples[a]}_{samples[b]}" for a, b in pairs]
    print("Analyzing pairs:")
    print(",".join(columns))

    totaln = len(events)
    for n in range(0, totaln, 50):
        print(f"[{n} / {totaln}] events analyzed...")

    own_ctx = ctx is None
    ctx = ctx if ctx is not None else Context(L.local_rank)
    header = "clusterID\t" + "\t".join(columns) + "\n"
    fractional = counts.dtype != np.int32
    if fractional and args.chi2:
        raise ValueError("pairwise --chi2 needs integer counts here (the chi-square kernel takes integer tables; the "
                         "reference would feed the fractional table to scipy.stats.chi2_contingency as it stands)")
    if fractional and totaln and pairs:
        # a table with non-integer cells: float sums in table order, truncated as scipy's int64 cast does; one rank
        if L.root:
            try:
                row_ptr, col = exclusion_csr(events, clusters)
                incl, excl = fractional_tables(ctx, counts, row_ptr, col)
                if hasattr(ctx, "fisher_pairs_dev"):
                    device_pipeline(ctx, incl, row_ptr, col, False, args.multiple_test_correction, events, header, args.output, excl)
                else:
                    parray = ctx.fisher_pairs(incl, excl)
                    if args.multiple_test_correction == "all":
                        parray = ctx.bh(parray.ravel()).reshape(parray.shape)
                    elif args.multiple_test_correction == "pairwise":
                        parray = ctx.bh_columns(parray)
                    textio.write_table(args.output, header, events, np.asarray(parray, dtype=np.float64), "repr")
            finally:
                if own_ctx:
                    ctx.close()
        elif own_ctx:
            ctx.close()
        return
    if L.world > 1 and totaln and pairs:
        # junction rows sharded over the ranks (distributed.pairwise_sharded); every rank formats its own
        # rows, rank 0 stitches the parts into the one output table
        from . import distributed
        try:
            row_ptr, col = exclusion_csr(events, clusters)
            out = distributed.pairwise_sharded(ctx, L.comm(ctx), np.ascontiguousarray(counts, dtype=np.int32), row_ptr, col,
                                               args.multiple_test_correction, test="chi2" if args.chi2 else "fisher")
        finally:
            if own_ctx:
                ctx.close()
        lo, hi = out["own"]
        textio.write_table(L.part(args.output), header if L.root else "", events[lo:hi],
                           np.asarray(out["p"], dtype=np.float64).reshape(hi - lo, len(pairs)), "repr")
        L.stitch(args.output)
        return
    if not L.root:
        if own_ctx:
            ctx.close()
        return                       # (empty inputs are not sharded: rank 0 alone)
    if totaln and pairs and hasattr(ctx, "fisher_pairs_dev"):
        try:
            row_ptr, col = exclusion_csr(events, clusters)
            device_pipeline(ctx, counts, row_ptr, col, args.chi2, args.multiple_test_correction, events, header, args.output)
        finally:
            if own_ctx:
                ctx.close()
        return
    try:        # host-array engine (the CPU test double) and empty inputs
        row_ptr, col = exclusion_csr(events, clusters)
        if totaln and pairs:
            excl = ctx.ps(counts, row_ptr, col, want_excl=True, want_ps=False)
            if args.chi2:
                parray, n_bad = ctx.chi2_pairs(counts, excl)
                if n_bad:
                    # scipy.stats.chi2_contingency raises on the first such table and the reference
                    # run dies with it (pairwise_fisher.py:167-179)
                    raise ValueError("The internally computed table of expected frequencies has a zero element "
                                     f"({n_bad} of {parray.size} sample-pair tables have an empty row or column)")
            else:
                parray = ctx.fisher_pairs(counts, excl)
            if args.multiple_test_correction == "all":
                parray = ctx.bh(parray.ravel()).reshape(parray.shape)
            elif args.multiple_test_correction == "pairwise":
                parray = ctx.bh_columns(parray)
        else:
            parray = np.zeros((totaln, len(pairs)))
    finally:
        if own_ctx:
            ctx.close()

    # str(numpy.float64) per cell (pairwise_fisher.py:195-200) through the library's formatter
    textio.write_table(args.output, header, events,
                       np.asarray(parray, dtype=np.float64).reshape(len(events), len(pairs)), "repr")


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    add_parser(p)
    run_with(p.parse_args())
