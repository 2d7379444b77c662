"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (numpy / pure-Python loops) of the splicedice hot path
(BASELINE.json north_star): overlap clustering, PS, rank-sum compare, pairwise
Fisher, BH-FDR.  Each function cites the reference file:line it follows
(paths relative to the reference checkout).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s cpu_baseline leg may import this module.

Pinning: the reference has no tests/golden vectors, so the oracle is
pinned against outputs of the reference itself, generated in the build container by
`tests/golden/make_golden.py` (which imports /root/reference) and committed under
`tests/golden/`; `tests/test_oracle_golden.py` holds the comparison.
One call is NOT pinned: `statsmodels...multipletests(method="fdr_bh")`
(compareSampleSets.py:235, pairwise_fisher.py:185,190) -- statsmodels is absent from
this image, so `bh_fdr` is "parity unpinned": restated from the published
Benjamini-Hochberg step-up definition and cross-checked against
`scipy.stats.false_discovery_control`.

Third-party arithmetic the reference calls (scipy.stats.ranksums / fisher_exact,
scipy 1.15.3 in this image; reference pins scipy==1.4.1 in requirements.txt:3) is
restated here (`ranksums_restated`, `fisher_exact_restated`) and pinned against the
installed scipy by KATs.
"""
import math
import numpy as np

# --------------------------------------------------------------------------------------
# quant / counts_to_ps
# --------------------------------------------------------------------------------------

def get_clusters(junctions):
    """SPLICEDICE.py:230-255 (twin: counts_to_ps.py:16-41), loop for loop.

    junctions: iterable of (chrom:str, left:int, right:int, strand:str).
    Returns dict junction -> list of overlapping junctions, in the reference's order:
    earlier-in-sweep overlaps most recent first, then later ones in sweep order.
    """
    chromosome = None
    strand = None
    clusters = {}
    potential = []
    for junction in sorted(junctions, key=lambda x: (x[0], x[3], x[1], x[2])):  # :237
        if junction[0] != chromosome or junction[3] != strand:                  # :240
            chromosome = junction[0]
            strand = junction[3]
            potential = []
        clusters[junction] = []
        new_potential = [junction]
        for prior in potential:
            if prior[2] >= junction[1]:                                         # :250 inclusive
                clusters[prior].append(junction)
                clusters[junction].append(prior)
                new_potential.append(prior)
        potential = new_potential
    return clusters


def cluster_csr(chrom_rank, left, right, strand):
    """Array form of get_clusters + junctionIndex (SPLICEDICE.py:96).

    Inputs are parallel arrays; chrom_rank is the dense rank of the chromosome name
    under Python string sort, strand 0='+' / 1='-' ('+' < '-').
    Returns (row_of[n] int32: output row of input junction i = rank in
    (chrom,left,right,strand) order; row_ptr[n+1] int64; col[nnz] int32: neighbour rows
    in the reference's list order).
    """
    n = len(chrom_rank)
    tuples = [(int(chrom_rank[i]), int(left[i]), int(right[i]), int(strand[i])) for i in range(n)]
    clusters = get_clusters(tuples)
    index = {j: i for i, j in enumerate(sorted(clusters))}                     # :96
    row_of = np.fromiter((index[t] for t in tuples), dtype=np.int32, count=n)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    cols = []
    for r, j in enumerate(sorted(clusters)):
        lst = clusters[j]
        row_ptr[r + 1] = row_ptr[r] + len(lst)
        cols.extend(index[o] for o in lst)
    return row_of, row_ptr, np.asarray(cols, dtype=np.int32)


def calculate_psi(counts, row_ptr, col, low=None):
    """SPLICEDICE.py:297-310.  counts [n,s] (any integer/float dtype, row order).

    Mirrors the dtypes: counts are float32 in the reference (:259), exclusions
    accumulate in float64 (:303-305), the quotient is float64 and is stored to a
    float32 matrix (:299,306); 0/0 -> NaN; `low` entries -> NaN (:307-309).
    Returns (psi float32 [n,s], excl int64 [n,s]).
    """
    counts32 = np.asarray(counts).astype(np.float32)
    n, s = counts32.shape
    psi = np.zeros((n, s), dtype=np.float32)
    excl_all = np.zeros((n, s), dtype=np.int64)
    with np.errstate(invalid="ignore", divide="ignore"):
        for r in range(n):
            inclusions = counts32[r, :]
            exclusions = np.zeros(s)
            for k in range(row_ptr[r], row_ptr[r + 1]):
                exclusions += counts32[col[k], :]
            psi[r, :] = inclusions / (inclusions + exclusions)
            excl_all[r, :] = exclusions.astype(np.int64)
    if low is not None:
        for r, c in low:
            psi[r, c] = np.nan
    return psi, excl_all


def calculate_psi_vectorised(counts, row_ptr, col):
    """Same arithmetic as calculate_psi without the Python loop (for larger test sizes).

    Integer exclusion sums via np.add.reduceat are exact; the quotient is
    float32(float64(incl) / float64(incl + excl)) exactly as SPLICEDICE.py:306.
    """
    counts = np.asarray(counts)
    n, s = counts.shape
    excl = np.zeros((n, s), dtype=np.int64)
    deg = np.diff(row_ptr)
    nz = np.flatnonzero(deg)
    if len(col):
        gathered = counts[col].astype(np.int64)
        excl[nz] = np.add.reduceat(gathered, row_ptr[nz], axis=0)
    incl = counts.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        psi = (incl / (incl + excl.astype(np.float64))).astype(np.float32)
    return psi, excl


def write_ps_values_f64(counts, row_ptr, col, n_out=None):
    """counts_to_ps.py:58-70 (writePsValues) on a float64 table: `exclusion = counts[j].copy()`, then
    `exclusion += counts[overlap]` per list entry in list order, `ps = counts[j] / exclusion`.
    counts [n_rows,s] float64 (fractional values allowed: the reference parses with dtype=float, :50);
    rows n_out.. are sources only.  Returns ps float64 [n_out,s]."""
    counts = np.asarray(counts, dtype=np.float64)
    n_rows, s = counts.shape
    n_out = n_rows if n_out is None else n_out
    ps = np.empty((n_out, s), dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        for r in range(n_out):
            exclusion = counts[r].copy()
            for k in range(row_ptr[r], row_ptr[r + 1]):
                exclusion += counts[col[k]]
            ps[r] = counts[r] / exclusion
    return ps


def quantize3(ps):
    """The `_allPS.tsv` text round trip: f'{x:.3f}' (SPLICEDICE.py:353) re-read as
    float32 (compareSampleSets.py:202)."""
    flat = np.asarray(ps, dtype=np.float32).ravel()
    out = np.array([np.float32(f"{float(x):.3f}") for x in flat], dtype=np.float32)
    return out.reshape(np.shape(ps))


def quantize3_fast(ps):
    """Closed form of quantize3: k = rint(float64(x) * 1000) is the '.3f' digit string
    (x*1000 is not always exact in float64, but ties of the decimal expansion are
    decided identically for float32 inputs -- checked exhaustively against quantize3
    in tests), value = float32(k / 1000.0)."""
    x = np.asarray(ps, dtype=np.float32).astype(np.float64)
    k = np.rint(x * 1000.0)
    return (k / 1000.0).astype(np.float32)


# --------------------------------------------------------------------------------------
# compare_sample_sets
# --------------------------------------------------------------------------------------

def rankdata_average(a):
    """scipy.stats.rankdata(method='average') restated."""
    a = np.asarray(a, dtype=np.float64)
    order = np.argsort(a, kind="mergesort")
    ranks = np.empty(a.size, dtype=np.float64)
    i = 0
    srt = a[order]
    while i < a.size:
        j = i
        while j + 1 < a.size and srt[j + 1] == srt[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return ranks


def ranksums_restated(x, y):
    """scipy.stats.ranksums two-sided (scipy/stats/_stats_py.py ranksums, 1.15.3):
    average ranks, no tie correction, no continuity correction; p = 2*ndtr(-|z|)."""
    x = np.asarray(x)
    y = np.asarray(y)
    n1, n2 = len(x), len(y)
    ranked = rankdata_average(np.concatenate((x, y)))
    s = np.sum(ranked[:n1])
    expected = n1 * (n1 + n2 + 1) / 2.0
    z = (s - expected) / np.sqrt(n1 * n2 * (n1 + n2 + 1) / 12.0)
    p = math.erfc(abs(z) / math.sqrt(2.0))   # == 2*ndtr(-|z|)
    return z, p


def compare_rows(matrix, g1_idx, g2_idx, use_scipy=True):
    """compareSampleSets.py:216-232, loop for loop.

    matrix float32 [n,s]; g1_idx/g2_idx column index arrays (table order, :96-102).
    Returns dict of un-compacted arrays: tested uint8[n], p f64[n], med1, med2, mean1,
    mean2, delta float32[n] (untested rows hold 0).
    """
    if use_scipy:
        from scipy.stats import ranksums
    matrix = np.asarray(matrix, dtype=np.float32)
    n = matrix.shape[0]
    tested = np.zeros(n, dtype=np.uint8)
    p = np.zeros(n, dtype=np.float64)
    z = np.zeros(n, dtype=np.float64)
    med1 = np.zeros(n, dtype=np.float32)
    med2 = np.zeros(n, dtype=np.float32)
    mean1 = np.zeros(n, dtype=np.float32)
    mean2 = np.zeros(n, dtype=np.float32)
    delta = np.zeros(n, dtype=np.float32)
    for r in range(n):
        event = matrix[r]
        d1, d2 = event[g1_idx], event[g2_idx]
        data1 = d1[np.invert(np.isnan(d1))]
        data2 = d2[np.invert(np.isnan(d2))]
        if len(data1) < 3 or len(data2) < 3:                                   # :223
            continue
        if use_scipy:
            zz, pval = ranksums(data1, data2)                                  # :226
        else:
            zz, pval = ranksums_restated(data1, data2)
        tested[r] = 1
        p[r] = pval
        z[r] = zz
        med1[r] = np.median(data1)
        med2[r] = np.median(data2)
        mean1[r] = np.mean(data1)
        mean2[r] = np.mean(data2)
        delta[r] = med1[r] - med2[r]
    return dict(tested=tested, p=p, z=z, med1=med1, med2=med2, mean1=mean1, mean2=mean2,
                delta=delta)


def bh_fdr(pvals):
    """Benjamini-Hochberg as statsmodels `multipletests(p, method='fdr_bh')[1]`
    computes it (compareSampleSets.py:235).  PARITY UNPINNED: statsmodels is not
    installable in this image; restated from its public definition:
    sort ascending, p_(i) / (i/m), reverse running minimum, clip to 1, unsort.
    """
    p = np.asarray(pvals, dtype=np.float64)
    m = p.size
    if m == 0:
        return p.copy()
    order = np.argsort(p, kind="mergesort")
    ps = p[order]
    ecdf = np.arange(1, m + 1) / float(m)
    raw = ps / ecdf
    corr = np.minimum.accumulate(raw[::-1])[::-1]
    corr[corr > 1] = 1
    out = np.empty_like(corr)
    out[order] = corr
    return out


# --------------------------------------------------------------------------------------
# pairwise
# --------------------------------------------------------------------------------------

def _log_hypergeom_pmf(k, M, n1, n):
    """log pmf of hypergeom(M=n1+n2 total, n1 'good', n draws) at k."""
    lg = math.lgamma
    return (lg(n1 + 1) - lg(k + 1) - lg(n1 - k + 1)
            + lg(M - n1 + 1) - lg(n - k + 1) - lg(M - n1 - n + k + 1)
            - (lg(M + 1) - lg(n + 1) - lg(M - n + 1)))


def fisher_exact_restated(a, b, c, d):
    """Two-sided p of scipy.stats.fisher_exact([[a,b],[c,d]]) (scipy/stats/_stats_py.py,
    1.15.3; call site pairwise_fisher.py:179).

    scipy's branches (mode test, cdf/sf + binary search for the opposite-tail boundary
    with slack 1+1e-14) amount to: sum pmf(k) over the support where
    pmf(k) <= pmf(a)*(1+1e-14), clipped at 1; any zero margin -> 1.0.
    Ratios pmf(k)/pmf(a) are walked by the exact recurrence from k=a so that ties are
    decided to ~1e-13, not at log-gamma accuracy.
    """
    a, b, c, d = int(a), int(b), int(c), int(d)
    n1, n2, n = a + b, c + d, a + c
    if n1 == 0 or n2 == 0 or n == 0 or (b + d) == 0:
        return 1.0
    M = n1 + n2
    lo = max(0, n - n2)
    hi = min(n1, n)
    pexact = math.exp(_log_hypergeom_pmf(a, M, n1, n))
    slack = 1.0 + 1e-12
    total = 1.0  # r_a
    # walk down from a
    r = 1.0
    k = a
    while k > lo:
        # pmf(k-1)/pmf(k) = k (n2-n+k) / ((n1-k+1)(n-k+1))
        r = r * (k * (n2 - n + k)) / ((n1 - k + 1) * (n - k + 1))
        k -= 1
        if r <= slack:
            total += r
    r = 1.0
    k = a
    while k < hi:
        # pmf(k+1)/pmf(k) = (n1-k)(n-k) / ((k+1)(n2-n+k+1))
        r = r * ((n1 - k) * (n - k)) / ((k + 1) * (n2 - n + k + 1))
        k += 1
        if r <= slack:
            total += r
    return min(pexact * total, 1.0)


def pairwise_exclusions(counts, row_ptr, col):
    """pairwise_fisher.py:158-160: exclusions = sum of count rows of the overlapping
    events present in the table (integer sums)."""
    _, excl = calculate_psi_vectorised(counts, row_ptr, col)
    return excl


def pair_list(s):
    """pairwise_fisher.py:142-147: (i,j), i<j, row-major."""
    return [(i, j) for i in range(s - 1) for j in range(i + 1, s)]


def fisher_pairs(incl, excl, use_scipy=True):
    """pairwise_fisher.py:154-180: p[n, s(s-1)/2] float64."""
    if use_scipy:
        from scipy.stats import fisher_exact
    incl = np.asarray(incl)
    excl = np.asarray(excl)
    n, s = incl.shape
    pairs = pair_list(s)
    out = np.empty((n, len(pairs)), dtype=np.float64)
    for r in range(n):
        for q, (i, j) in enumerate(pairs):
            if use_scipy:
                table = [[incl[r, i], incl[r, j]], [excl[r, i], excl[r, j]]]
                out[r, q] = fisher_exact(table)[1]
            else:
                out[r, q] = fisher_exact_restated(incl[r, i], incl[r, j], excl[r, i], excl[r, j])
    return out


def chi2_yates_restated(a, b, c, d):
    """p of scipy.stats.chi2_contingency([[a,b],[c,d]]) (contingency.py, 1.15.3; pairwise --chi2,
    pairwise_fisher.py:133-136): Yates-corrected Pearson chi-square, dof 1, p = chdtrc(1, chi2).
    Raises ValueError on a zero expected frequency, as scipy does."""
    obs = np.array([[a, b], [c, d]], dtype=np.float64)
    tot = obs.sum()
    exp = np.outer(obs.sum(axis=1), obs.sum(axis=0)) / tot if tot > 0 else np.zeros((2, 2))
    if (exp == 0).any():
        raise ValueError("The internally computed table of expected frequencies has a zero element")
    diff = exp - obs
    obs = obs + np.minimum(0.5, np.abs(diff)) * np.sign(diff)
    stat = (((obs - exp) ** 2) / exp).sum()
    return math.erfc(math.sqrt(0.5 * stat))


def chi2_pairs(incl, excl, use_scipy=True):
    """pairwise --chi2: p[n, s(s-1)/2]; ValueError as scipy raises it."""
    if use_scipy:
        from scipy.stats import chi2_contingency
    incl = np.asarray(incl)
    excl = np.asarray(excl)
    n, s = incl.shape
    pairs = pair_list(s)
    out = np.empty((n, len(pairs)), dtype=np.float64)
    for r in range(n):
        for q, (i, j) in enumerate(pairs):
            if use_scipy:
                out[r, q] = chi2_contingency([[incl[r, i], incl[r, j]], [excl[r, i], excl[r, j]]])[1]
            else:
                out[r, q] = chi2_yates_restated(incl[r, i], incl[r, j], excl[r, i], excl[r, j])
    return out


def bh_columns(p):
    """pairwise_fisher.py:187-191: BH down each pair column."""
    p = np.array(p, dtype=np.float64)
    for i in range(p.shape[1]):
        p[:, i] = bh_fdr(p[:, i])
    return p


# --------------------------------------------------------------------------------------
# similarity
# --------------------------------------------------------------------------------------

def similarity_scores(ps, mid, sign):
    """similarity.py:24-47 on arrays: ps float64 [n, s] (NaN = "nan" field), per-row midpoint and
    sign of delta (0 = event not in the comparison).  Loop for loop as the reference."""
    ps = np.asarray(ps, dtype=np.float64)
    n, s = ps.shape
    scores, counts = [0] * s, [0] * s
    for r in range(n):
        if sign[r] < 0:
            for i in range(s):
                if not np.isnan(ps[r, i]):
                    counts[i] += 1
                    if float(ps[r, i]) < mid[r]:
                        scores[i] += 1
        elif sign[r] > 0:
            for i in range(s):
                if not np.isnan(ps[r, i]):
                    counts[i] += 1
                    if float(ps[r, i]) > mid[r]:
                        scores[i] += 1
    return np.array(scores, dtype=np.int64), np.array(counts, dtype=np.int64)


# --------------------------------------------------------------------------------------
# findOutliers
# --------------------------------------------------------------------------------------

def find_outlier_lines(rows, cols, matrix, samples, null, cutoff):
    """findOutliers.py:117-152 row by row -> list of output lines (without newline)."""
    null_idx = np.argwhere(np.isin(cols, null))[:, 0]
    out_idx = np.argwhere(np.isin(cols, samples))[:, 0]
    lines = []
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for e, row in enumerate(matrix):
                null_vals = row[null_idx]
                if np.sum(np.isnan(null_vals)) / len(null_vals) > 0.2:
                    continue
                std = np.nanstd(null_vals)
                if std < 0.001:
                    continue
                mean = np.nanmean(null_vals)
                vals = row[out_idx]
                z = (vals - mean) / std
                for pos, x in enumerate(z):
                    if x >= cutoff and abs(x - mean) >= cutoff:
                        lines.append("\t".join(str(v) for v in (rows[e], cols[out_idx[pos]], x, vals[pos], mean, std)))
    return lines
