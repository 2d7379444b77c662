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


def test_pairwise_config4_shard_full_shape(ctx):
    """BASELINE config 4, one GPU's shard: 25 000 junctions x 200 samples -> 19 900 pair columns, device resident.
    Pair-index inversion up to the last column, column-wise BH properties, sampled cells against scipy."""
    from scipy.stats import fisher_exact
    n, s = 25_000, 200
    pairs = s * (s - 1) // 2
    cr, l, r, st = synth.make_junctions(n, 4)
    row_of, row_ptr, col = ctx.cluster(cr, l, r, st)
    counts_in = synth.make_counts(n, s, 40)
    counts = np.zeros_like(counts_in)
    counts[row_of] = counts_in
    d_counts, d_rp, d_col = ctx.to_device(counts), ctx.to_device(row_ptr), ctx.to_device(col)
    d_excl, d_p = ctx.empty((n, s), np.int64), ctx.empty((n, pairs), np.float64)
    ctx.ps_dev(d_counts, d_rp, d_col, d_excl, None)
    ctx.fisher_pairs_dev(d_counts, d_excl, d_p)
    excl = d_excl.to_host()
    _, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    assert np.array_equal(excl, want_excl)
    rows = [0, 1, 12_345, n - 1]
    raw = {rr: d_p.offset(rr * pairs, (pairs,)).to_host() for rr in rows}
    def pair_index(i, j):
        return i * s - i * (i + 1) // 2 + (j - i - 1)
    assert pair_index(s - 2, s - 1) == pairs - 1
    for rr in rows:
        assert ((raw[rr] > 0) & (raw[rr] <= 1)).all()
        for i, j in [(0, 1), (0, s - 1), (1, 2), (57, 133), (s - 3, s - 1), (s - 2, s - 1)]:     # first ... last column
            w = fisher_exact([[counts[rr, i], counts[rr, j]], [excl[rr, i], excl[rr, j]]])[1]
            assert abs(raw[rr][pair_index(i, j)] - w) <= 1e-9 * w, (rr, i, j)
    # column-wise BH on the whole [25 000, 19 900] table, in place
    cols_chk = [0, 1, 9_950, pairs - 2, pairs - 1]
    def column(c):
        out = ctx.empty((n,), np.float64)
        ctx.copy2d_dev(out.ptr, 8, d_p.ptr + c * 8, pairs * 8, 8, n)
        return out.to_host()
    before = {c: column(c) for c in cols_chk}
    ctx.bh_columns_dev(d_p)
    for c in cols_chk:
        q, p = column(c), before[c]
        order = np.argsort(p, kind="stable")
        assert (np.diff(q[order]) >= -1e-18).all() and (q >= p * (1 - 1e-15)).all() and (q <= 1).all()
        np.testing.assert_allclose(q, O.bh_fdr(p), rtol=1e-12, atol=0)


def test_e2e_config5_shard_full_shape(ctx):
    """BASELINE config 5, one GPU's shard: 625 000 junctions x 1000 samples, 500 v 500, device resident:
    cluster -> PS (quantise fused) -> rank-sum.  Group-swap antisymmetry on every row, sampled rows vs the oracle."""
    n, s, blk = 625_000, 1000, 125_000
    cr, l, r, st = synth.make_junctions(n, 5)
    d_j = [ctx.to_device(x) for x in (cr, l, r, st)]
    d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_col, nnz = ctx.cluster_dev(*d_j, d_row_of, d_rp)
    block = synth.make_counts(blk, s, 50)
    d_counts = ctx.empty((n, s), np.int32)
    for a in range(0, n, blk):
        d_counts.offset(a * s, (blk, s)).upload(block)
    d_ps = ctx.empty((n, s), np.float32)
    ctx.set_param("ps.quantize3", 1)
    try:
        ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
    finally:
        ctx.set_param("ps.quantize3", 0)
    g1, g2 = np.arange(0, 500, dtype=np.int32), np.arange(500, 1000, dtype=np.int32)
    d_g1, d_g2 = ctx.to_device(g1), ctx.to_device(g2)
    def run(a, b):
        out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64),
                   med1=ctx.empty(n, np.float32), med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32),
                   mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
        ctx.ranksum_dev(d_ps, a, b, out)
        return {k: v.to_host() for k, v in out.items()}
    fwd, rev = run(d_g1, d_g2), run(d_g2, d_g1)
    t = fwd["tested"].astype(bool)
    assert np.array_equal(fwd["tested"], rev["tested"]) and t.mean() > 0.9
    assert np.array_equal(fwd["z"][t], -rev["z"][t]) and np.array_equal(fwd["p"][t], rev["p"][t])
    assert np.array_equal(fwd["med1"], rev["med2"]) and np.array_equal(fwd["mean2"], rev["mean1"])
    assert np.array_equal(fwd["delta"][t], -rev["delta"][t])
    # sampled rows against the oracle, on the device's own PS rows (rows 0.., a block seam, the last rows)
    for lo in (0, blk - 150, n - 300):
        ps = d_ps.offset(lo * s, (300, s)).to_host()
        want = O.compare_rows(ps, g1, g2)
        tt = want["tested"].astype(bool)
        sl = slice(lo, lo + 300)
        assert np.array_equal(fwd["tested"][sl], want["tested"]) and np.array_equal(fwd["z"][sl][tt], want["z"][tt])
        for k in ("med1", "med2", "mean1", "mean2", "delta"):
            assert np.array_equal(fwd[k][sl][tt], want[k][tt]), k
        np.testing.assert_allclose(fwd["p"][sl][tt], want["p"][tt], rtol=1e-9, atol=0)
    # and the PS rows themselves where every neighbour row is at hand: first rows of the table
    rp = d_rp.to_host()[:4001]
    cl = d_col.to_host()[: int(rp[-1])]
    inside = np.array([(cl[rp[i]:rp[i + 1]] < 4000).all() for i in range(300)])
    want_ps, _ = O.calculate_psi_vectorised(block[:4000], rp, np.minimum(cl, 3999))
    got = d_ps.offset(0, (300, s)).to_host()
    assert np.array_equal(got[inside], O.quantize3_fast(want_ps[:300])[inside], equal_nan=True)
