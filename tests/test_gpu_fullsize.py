"""GPU, BASELINE.json sizes: size-independent properties of the HIP path (the oracle's Python loops
cannot run at 1M rows in test time; a row sample is still compared with it)."""
import numpy as np
import pytest

from oracle import oracle_np as O
from splicedice_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(ctx):
    """config 2: 1M junctions x 100 samples"""
    n, s = 1_000_000, 100
    cr, left, right, strand = synth.make_junctions(n, 2)
    row_of, row_ptr, col = ctx.cluster(cr, left, right, strand)
    rng = np.random.default_rng(5)
    counts = rng.negative_binomial(2, 2 / 32.0, size=(n, s)).astype(np.int32)
    counts[rng.random((n, s)) < 0.2] = 0
    return dict(n=n, s=s, cr=cr, left=left, right=right, strand=strand, row_of=row_of, row_ptr=row_ptr, col=col,
                counts=counts)


def test_cluster_properties_1m(c2):
    n, row_of, row_ptr, col = c2["n"], c2["row_of"], c2["row_ptr"], c2["col"]
    # row_of is a permutation and rows are in (chrom, left, right, strand) order
    inv = np.empty(n, np.int64)
    inv[row_of] = np.arange(n)
    assert np.array_equal(np.sort(row_of), np.arange(n))
    cr, l, r, st = (c2[k][inv].astype(np.int64) for k in ("cr", "left", "right", "strand"))
    key = np.stack([cr, l, r, st], axis=1)
    assert np.array_equal(np.lexsort((st, r, l, cr)), np.arange(n))        # sortedness
    # every listed neighbour overlaps (inclusive), same chrom and strand, never itself
    rows = np.repeat(np.arange(n), np.diff(row_ptr))
    assert (cr[rows] == cr[col]).all() and (st[rows] == st[col]).all() and (rows != col).all()
    assert (r[rows] >= l[col]).all() and (r[col] >= l[rows]).all()
    # symmetry: the edge multiset equals its transpose
    fwd = rows * n + col
    bwd = col.astype(np.int64) * n + rows
    assert np.array_equal(np.sort(fwd), np.sort(bwd))
    # completeness: nnz equals the closed-form count 2 * sum_p (ub_p - p - 1) in sweep order
    order = np.lexsort((r, l, st, cr))
    seg = cr[order] * 2 + st[order]
    ckl = seg * (1 << 32) + l[order]
    ub = np.searchsorted(ckl, seg * (1 << 32) + r[order], side="right")
    assert col.size == 2 * int((ub - np.arange(n) - 1).sum())
    # list order on a sample of rows: identical to the reference sweep restated by the oracle
    sample = np.flatnonzero((cr == 3))[:4000]
    lo, hi = sample[0], sample[-1] + 1
    # rows [lo, hi) are a contiguous part of one chromosome; neighbours may fall outside -> compare by coordinates
    _, rp_o, col_o = O.cluster_csr(cr[lo:hi], l[lo:hi], r[lo:hi], st[lo:hi])
    for k in range(200, hi - lo - 200, 37):   # interior rows: every neighbour is inside the slice (reach ~25 rows)
        want = col_o[rp_o[k]:rp_o[k + 1]] + lo
        assert np.array_equal(col[row_ptr[lo + k]:row_ptr[lo + k + 1]], want)


def test_ps_properties_1m(ctx, c2):
    n, s, row_ptr, col, counts = c2["n"], c2["s"], c2["row_ptr"], c2["col"], c2["counts"]
    ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
    # checksum of checksums: sum_r excl[r, :] == sum_j deg_j * counts[j, :]  (the CSR is symmetric)
    deg = np.diff(row_ptr)
    assert np.array_equal(excl.sum(axis=0), (counts.astype(np.int64) * deg[:, None]).sum(axis=0))
    # PS reconstructs from the sums exactly as the reference arithmetic prescribes
    with np.errstate(invalid="ignore", divide="ignore"):
        want = (counts.astype(np.float64) / (counts.astype(np.float64) + excl.astype(np.float64))).astype(np.float32)
    assert np.array_equal(ps, want, equal_nan=True)
    # a row sample against the loop-for-loop oracle
    rows = np.arange(123_000, 123_400)
    for r in rows[::17]:
        e = np.zeros(s)
        for k in range(row_ptr[r], row_ptr[r + 1]):
            e += counts[col[k]].astype(np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            assert np.array_equal(ps[r], (counts[r].astype(np.float32) / (counts[r].astype(np.float32) + e)).astype(np.float32),
                                  equal_nan=True)
    # idempotence of the text quantisation
    q = ctx.quantize3(ps[:200_000])
    assert np.array_equal(ctx.quantize3(q), q, equal_nan=True)


def test_ranksum_and_bh_properties_1m(ctx):
    n, s = 1_000_000, 100
    ps = synth.make_ps_matrix(n, s, 3)
    g1, g2 = np.arange(0, 50, dtype=np.int32), np.arange(50, 100, dtype=np.int32)
    a = ctx.ranksum(ps, g1, g2)
    b = ctx.ranksum(ps, g2, g1)                  # swapping the groups negates z and keeps p
    t = a["tested"].astype(bool)
    assert np.array_equal(a["tested"], b["tested"]) and t.mean() > 0.95
    assert np.array_equal(a["z"][t], -b["z"][t]) and np.array_equal(a["p"][t], b["p"][t])
    assert np.array_equal(a["med1"], b["med2"]) and np.array_equal(a["mean1"], b["mean2"])
    assert (a["p"][t] > 0).all() and (a["p"][t] <= 1).all()
    perm = np.random.default_rng(1).permutation(50).astype(np.int32)     # order inside a group is irrelevant
    c = ctx.ranksum(ps[:100_000], np.sort(g1[perm]), g2)
    assert np.array_equal(c["p"], a["p"][:100_000])
    sample = np.arange(500_000, 500_300)
    want = O.compare_rows(ps[sample], g1, g2)
    assert np.array_equal(a["z"][sample], want["z"]) and np.array_equal(a["mean1"][sample], want["mean1"])
    # BH: monotone in p, q >= p, q <= 1, idempotent ordering
    p = a["p"][t]
    q = ctx.bh(p)
    order = np.argsort(p, kind="stable")
    assert (np.diff(q[order]) >= 0).all() and (q >= p).all() and (q <= 1).all()
    assert np.isclose(q[order][-1], p[order][-1])
    np.testing.assert_allclose(q[order][:1000], O.bh_fdr(p)[order][:1000], rtol=1e-12)


def test_fisher_properties(ctx):
    rng = np.random.default_rng(2)
    t = rng.integers(0, 400, size=(200_000, 4))
    p = ctx.fisher_tables(t)
    assert ((p > 0) & (p <= 1)).all()
    # invariances of the 2x2 table: transpose, row swap, column swap
    for perm in ([0, 2, 1, 3], [2, 3, 0, 1], [1, 0, 3, 2]):
        np.testing.assert_allclose(ctx.fisher_tables(t[:, perm]), p, rtol=1e-9)
    from scipy.stats import fisher_exact
    for i in range(0, 200_000, 9973):
        w = fisher_exact(t[i].reshape(2, 2))[1]
        assert abs(p[i] - w) <= 1e-9 * w
