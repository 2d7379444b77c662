"""GPU, BASELINE.json configs 4 and 5 at their FULL shape on ONE MI355X, and the headline 2 M x 500, device resident.

Every table here crosses 2^32 bytes and (configs 4, 5) 2^31 elements, so each index / offset path of the kernels
above 4 GiB is under test.  The oracle's Python loops cannot run at these sizes: the checks are the domain's
size-independent properties on ALL rows plus oracle comparisons on row windows at the start of the table, at every
2^32-byte boundary of it, and at its end.  Reference shapes: SPLICEDICE.py:297-310 (PS), compareSampleSets.py:216-235
(rank-sum + BH), pairwise_fisher.py:142-193 (all pairs x all junctions, BH per pair column).
"""
import numpy as np
import pytest

from oracle import oracle_np as O
from splicedice_amd import synth

pytestmark = pytest.mark.gpu


def _fill_repeated(d_table, block, n, s):
    """the table is one seeded block repeated down the device matrix (the host never holds the whole table)"""
    blk = block.shape[0]
    for a in range(0, n, blk):
        rows = min(blk, n - a)
        d_table.offset(a * s, (rows, s)).upload(block[:rows])


def _ps_window(d_counts, rp, col, lo, hi, s, n):
    """oracle PS of rows [lo, hi) from the device's own count rows; -> (ps, inside) where `inside` marks the rows whose
    neighbours all lie in the fetched window [lo - 64, hi + 64)"""
    w0, w1 = max(0, lo - 64), min(n, hi + 64)
    counts = d_counts.offset(w0 * s, (w1 - w0, s)).to_host()
    k0, k1 = int(rp[w0]), int(rp[w1])
    seg = col[k0:k1].astype(np.int64) - w0
    rp_w = (rp[w0:w1 + 1] - k0).astype(np.int64)
    ok_entry = (seg >= 0) & (seg < w1 - w0)
    bad_rows = np.unique(np.repeat(np.arange(w1 - w0), np.diff(rp_w))[~ok_entry])
    inside = np.ones(w1 - w0, bool)
    inside[bad_rows] = False
    ps, _ = O.calculate_psi_vectorised(counts, rp_w, np.clip(seg, 0, w1 - w0 - 1).astype(np.int32))
    return ps[lo - w0:hi - w0], inside[lo - w0:hi - w0]


def _boundary_windows(n, row_bytes, rows=256):
    """row windows at the start, around every 2^32-byte boundary of a table with `row_bytes` per row, and at the end"""
    out = [0]
    k = 1
    while k * (1 << 32) < n * row_bytes:
        out.append(min(n - rows, max(0, k * (1 << 32) // row_bytes - rows // 2)))
        k += 1
    out.append(n - rows)
    return out


def test_config5_full_5m_x_1000(ctx):
    """BASELINE config 5 on one GPU: 5 M junctions x 1000 samples (20 GB of counts, 20 GB of PS), 500 v 500:
    cluster -> PS ('.3f' round trip fused) -> rank-sum -> BH over the tested rows."""
    n, s, blk = 5_000_000, 1000, 50_000
    cr, l, r, st = synth.make_junctions(n, 55)
    d_j = [ctx.to_device(x) for x in (cr, l, r, st)]
    d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_col, nnz = ctx.cluster_dev(*d_j, d_row_of, d_rp)
    assert nnz > 4 * n
    block = synth.make_counts(blk, s, 56)
    d_counts, d_ps = ctx.empty((n, s), np.int32), ctx.empty((n, s), np.float32)
    assert d_counts.nbytes > 4 * (1 << 32) and n * s > (1 << 32)
    keys = ("tested", "p", "z", "med1", "med2", "mean1", "mean2", "delta")
    try:
        _fill_repeated(d_counts, block, n, s)
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
            return out

        d_fwd = run(d_g1, d_g2)
        d_q = ctx.empty(n, np.float64)
        ctx.bh_masked_dev(d_fwd["p"], d_fwd["tested"], d_q)
        fwd = {k: d_fwd[k].to_host() for k in keys}
        rev = {k: v.to_host() for k, v in run(d_g2, d_g1).items()}
        q = d_q.to_host()
        # group-swap antisymmetry on ALL rows
        t = fwd["tested"].astype(bool)
        assert np.array_equal(fwd["tested"], rev["tested"]) and t.mean() > 0.9
        assert np.array_equal(fwd["z"][t], -rev["z"][t]) and np.array_equal(fwd["p"][t], rev["p"][t])
        assert np.array_equal(fwd["med1"], rev["med2"]) and np.array_equal(fwd["mean2"], rev["mean1"])
        assert np.array_equal(fwd["delta"][t], -rev["delta"][t])
        assert ((fwd["p"][t] > 0) & (fwd["p"][t] <= 1)).all()
        # BH over the tested rows (compareSampleSets.py:235): the whole 5 M vector against the restated definition
        want_q = O.bh_fdr(fwd["p"][t])
        np.testing.assert_allclose(q[t], want_q, rtol=1e-12, atol=0)
        assert (q[~t] == 0).all()
        # row windows: PS rows against the oracle on the device's own counts, rank-sum rows on the device's own PS rows
        rp, col = d_rp.to_host(), d_col.to_host()
        assert rp[-1] == nnz
        wins = _boundary_windows(n, s * 4)
        assert len(wins) == 6                       # start, four 2^32-byte boundaries, end
        for lo in wins:
            hi = lo + 256
            want_ps, inside = _ps_window(d_counts, rp, col, lo, hi, s, n)
            got_ps = d_ps.offset(lo * s, (256, s)).to_host()
            assert inside.sum() > 128
            assert np.array_equal(got_ps[inside], O.quantize3_fast(want_ps)[inside], equal_nan=True), lo
            want = O.compare_rows(got_ps, g1, g2)
            tt = want["tested"].astype(bool)
            sl = slice(lo, hi)
            assert np.array_equal(fwd["tested"][sl], want["tested"]) and np.array_equal(fwd["z"][sl][tt], want["z"][tt])
            for k in ("med1", "med2", "mean1", "mean2", "delta"):
                assert np.array_equal(fwd[k][sl][tt], want[k][tt]), (lo, k)
            np.testing.assert_allclose(fwd["p"][sl][tt], want["p"][tt], rtol=1e-9, atol=0)
    finally:
        for d in (d_counts, d_ps):
            d.free()
        ctx.trim()


def test_config4_full_200k_x_200(ctx):
    """BASELINE config 4 on one GPU: 200 000 junctions x 200 samples -> 19 900 pair columns (3.98e9 p-values, 31.8 GB):
    exclusion sums + Fisher + BH down every pair column (pairwise_fisher.py:154-193)."""
    from scipy.stats import fisher_exact
    n, s = 200_000, 200
    pairs = s * (s - 1) // 2
    assert n * pairs > (1 << 31)
    cr, l, r, st = synth.make_junctions(n, 44)
    row_of, row_ptr, col = ctx.cluster(cr, l, r, st)
    counts_in = synth.make_counts(n, s, 45)
    counts = np.zeros_like(counts_in)
    counts[row_of] = counts_in
    d_counts, d_rp, d_col = ctx.to_device(counts), ctx.to_device(row_ptr), ctx.to_device(col)
    d_excl, d_p = ctx.empty((n, s), np.int64), ctx.empty((n, pairs), np.float64)
    try:
        ctx.ps_dev(d_counts, d_rp, d_col, d_excl, None)
        ctx.fisher_pairs_dev(d_counts, d_excl, d_p)
        excl = d_excl.to_host()
        _, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
        assert np.array_equal(excl, want_excl)

        def pair_index(i, j):
            return i * s - i * (i + 1) // 2 + (j - i - 1)
        # rows at the start, on both sides of every 2^32-byte boundary of the p-value table, and at the end
        row_bytes = pairs * 8
        rows = sorted({0, 1, n - 1} | {k * (1 << 32) // row_bytes + d for k in range(1, (n * row_bytes >> 32) + 1) for d in (0, 1)})
        assert len(rows) >= 15
        rng = np.random.default_rng(3)
        for rr in rows:
            raw = d_p.offset(rr * pairs, (pairs,)).to_host()
            assert ((raw > 0) & (raw <= 1)).all(), rr
            picks = [(0, 1), (0, s - 1), (s - 2, s - 1)] + [tuple(sorted(rng.choice(s, 2, replace=False))) for _ in range(3)]
            for i, j in picks:
                w = fisher_exact([[counts[rr, i], counts[rr, j]], [excl[rr, i], excl[rr, j]]])[1]
                assert abs(raw[pair_index(i, j)] - w) <= 1e-9 * w, (rr, i, j)

        def column(c):
            out = ctx.empty((n,), np.float64)
            ctx.copy2d_dev(out.ptr, 8, d_p.ptr + c * 8, pairs * 8, 8, n)
            return out.to_host()
        cols_chk = [0, 1, 7_777, pairs // 2, pairs - 2, pairs - 1]
        before = {c: column(c) for c in cols_chk}
        first_row = d_p.offset(0, (pairs,)).to_host()
        ctx.bh_columns_dev(d_p)
        for c in cols_chk:
            qc, pc = column(c), before[c]
            order = np.argsort(pc, kind="stable")
            assert (np.diff(qc[order]) >= -1e-18).all() and (qc >= pc * (1 - 1e-15)).all() and (qc <= 1).all()
            np.testing.assert_allclose(qc, O.bh_fdr(pc), rtol=1e-12, atol=0)
        # every column was corrected: no value of the first and the last row is below its p-value
        assert (d_p.offset(0, (pairs,)).to_host() >= first_row * (1 - 1e-15)).all()
    finally:
        d_p.free()
        d_excl.free()
        ctx.trim()


def test_headline_2m_x_500_properties(ctx):
    """the bench.py headline shape: 2 M junctions x 500 samples, cluster -> PS with exclusion sums, verified in row
    slabs: checksum of checksums (sum_r excl[r] == sum_j deg_j counts[j]: the CSR is symmetric) and the PS
    reconstruction from the sums (SPLICEDICE.py:306), over the first, the last and every fourth slab of 50 000 rows."""
    n, s, blk = 2_000_000, 500, 100_000
    cr, l, r, st = synth.make_junctions(n, 66)
    d_j = [ctx.to_device(x) for x in (cr, l, r, st)]
    d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_col, nnz = ctx.cluster_dev(*d_j, d_row_of, d_rp)
    block = synth.make_counts(blk, s, 67)
    d_counts, d_ps, d_excl = ctx.empty((n, s), np.int32), ctx.empty((n, s), np.float32), ctx.empty((n, s), np.int64)
    try:
        _fill_repeated(d_counts, block, n, s)
        ctx.ps_dev(d_counts, d_rp, d_col, d_excl, d_ps)
        only_ps = ctx.empty((n, s), np.float32)
        ctx.ps_dev(d_counts, d_rp, d_col, None, only_ps)           # the bench's instantiation (no exclusion sums)
        rp = d_rp.to_host()
        deg = np.diff(rp)
        slab = 50_000
        excl_sum = np.zeros(s, np.int64)
        deg_sum = np.zeros(s, np.int64)
        for a in range(0, n, slab):
            excl = d_excl.offset(a * s, (slab, s)).to_host()
            excl_sum += excl.sum(axis=0)
            counts = block[(a % blk):(a % blk) + slab]
            deg_sum += (counts.astype(np.int64) * deg[a:a + slab, None]).sum(axis=0)
            if a == 0 or a == n - slab or (a // slab) % 4 == 1:
                ps = d_ps.offset(a * s, (slab, s)).to_host()
                with np.errstate(invalid="ignore", divide="ignore"):
                    c64 = counts.astype(np.float64)
                    want = (c64 / (c64 + excl.astype(np.float64))).astype(np.float32)
                assert np.array_equal(ps, want, equal_nan=True), a
                assert np.array_equal(only_ps.offset(a * s, (slab, s)).to_host(), ps, equal_nan=True), a
        assert np.array_equal(excl_sum, deg_sum)
        # oracle rows at the start and the end of the table
        col = d_col.to_host()
        for lo in (0, n - 256):
            want_ps, inside = _ps_window(d_counts, rp, col, lo, lo + 256, s, n)
            got = d_ps.offset(lo * s, (256, s)).to_host()
            assert inside.sum() > 128 and np.array_equal(got[inside], want_ps[inside], equal_nan=True)
    finally:
        for d in (d_counts, d_ps, d_excl):
            d.free()
        ctx.trim()
