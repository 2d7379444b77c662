"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden
fixtures generated from the reference.  Bars: clusters / integer sums / PS / medians / means
bit-exact; z bit-exact; p-values within 1e-6 relative (north_star), tolerance written below.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle_np as O
from splicedice_amd import synth

pytestmark = pytest.mark.gpu

P_RTOL = 1e-6          # north_star tolerance for p-values
P_RTOL_TIGHT = 1e-9    # what the kernels actually reach on the KATs


def _golden_arrays(golden_dir, tag):
    return np.load(os.path.join(golden_dir, "arrays", f"cluster_psi_{tag}.npz"))


# ------------------------------------------------------------------------------ clustering
@pytest.mark.parametrize("tag", ["a", "b", "c", "dense"])
def test_cluster_golden(ctx, golden_dir, tag):
    z = _golden_arrays(golden_dir, tag)
    row_of, row_ptr, col = ctx.cluster(z["chrom_rank"], z["left"], z["right"], z["strand"])
    assert np.array_equal(row_of, z["row_of"])
    assert np.array_equal(row_ptr, z["row_ptr"])
    assert np.array_equal(col, z["col"])


@pytest.mark.parametrize("n,seed,kw", [
    (1, 1, {}), (2, 2, {}), (63, 3, {}), (64, 4, {}), (65, 5, {}), (1000, 6, {}),
    (20000, 7, {}), (20000, 8, dict(n_chrom=1)), (30000, 9, dict(n_chrom=300, gene_spacing=3000)),
    (5000, 10, dict(n_chrom=2, gene_spacing=50, len_span=100000)),   # very dense: long backward walks
])
def test_cluster_vs_oracle(ctx, n, seed, kw):
    cr, left, right, strand = synth.make_junctions(n, seed, **kw)
    want = O.cluster_csr(cr, left, right, strand)
    got = ctx.cluster(cr, left, right, strand)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("generic", [0, 1, 2])
def test_cluster_paths_and_wide_keys(ctx, generic):
    """fast path (0) vs the radix paths of cluster.hip: generic two-sort (1) and packed single-sort (2);
    coordinates near 2^31 and chrom ranks up to 2^20 force wide fields (the packed path must refuse
    keys that do not fit)."""
    rng = np.random.default_rng(77)
    n = 4000
    cr = rng.choice(np.array([0, 1, 7, 1 << 20, (1 << 20) + 1], np.int32), size=n)
    left = rng.integers(0, 40, size=n).astype(np.int64) * 5000 + rng.choice([0, 2_000_000_000], size=n)
    right = np.minimum(left + rng.integers(0, 30000, size=n), 2 ** 31 - 1)
    strand = rng.integers(0, 2, size=n).astype(np.int8)
    key = np.stack([cr, left, right, strand], axis=1)
    key = np.unique(key, axis=0)
    rng.shuffle(key)
    cr, left, right, strand = key[:, 0].astype(np.int32), key[:, 1].astype(np.int32), key[:, 2].astype(np.int32), key[:, 3].astype(np.int8)
    want = O.cluster_csr(cr, left, right, strand)
    ctx.set_param("cluster.generic", int(generic == 1))
    ctx.set_param("cluster.legacy", int(generic == 2))
    try:
        got = ctx.cluster(cr, left, right, strand)
        cr2, l2, r2, s2 = synth.make_junctions(9000, 31, n_chrom=7)
        got2 = ctx.cluster(cr2, l2, r2, s2)
    finally:
        ctx.set_param("cluster.generic", 0)
        ctx.set_param("cluster.legacy", 0)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    for g, w in zip(got2, O.cluster_csr(cr2, l2, r2, s2)):
        assert np.array_equal(g, w)


def test_cluster_touching_and_nested(ctx):
    # (100,200)/(200,300) touch -> neighbours; (201,250) does not overlap (100,200)
    cr = np.zeros(6, np.int32)
    left = np.array([100, 200, 201, 100, 100, 150], np.int32)
    right = np.array([200, 300, 250, 200, 1000, 160], np.int32)
    strand = np.array([0, 0, 0, 1, 0, 0], np.int8)
    want = O.cluster_csr(cr, left, right, strand)
    got = ctx.cluster(cr, left, right, strand)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("n,seed,kw,lds_cap", [
    (2049, 41, {}, 0),                                   # two buckets
    (70000, 42, {}, 0),                                  # 35 buckets, many tiles of the neighbour kernel
    (70000, 43, dict(n_chrom=1), 0),
    (40000, 44, dict(n_chrom=900, gene_spacing=2500), 0),
    (30000, 45, {}, 64),                                 # every bucket beyond the LDS capacity: in-place HBM sort
    (3000, 46, dict(n_chrom=2, gene_spacing=40, len_span=150000), 0),   # degree ~ hundreds: list longer than 16 n
    (300_000, 47, {}, 0),
])
def test_cluster_fast_path_buckets(ctx, n, seed, kw, lds_cap):
    """sample sort + tiled neighbour lists of cluster_fast.hip across bucket / tile boundaries"""
    cr, left, right, strand = synth.make_junctions(n, seed, **kw)
    if n <= 70000:
        want = O.cluster_csr(cr, left, right, strand)
    else:                                                # the oracle's Python sweep is too slow here: radix path
        ctx.set_param("cluster.legacy", 1)
        try:
            want = ctx.cluster(cr, left, right, strand)
        finally:
            ctx.set_param("cluster.legacy", 0)
    ctx.set_param("cluster.lds_cap", lds_cap)
    try:
        got = ctx.cluster(cr, left, right, strand)
    finally:
        ctx.set_param("cluster.lds_cap", 0)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_cluster_sorted_and_reversed_input(ctx):
    """input already in row order (what `quant` hands over after the junction union) and reversed"""
    cr, left, right, strand = synth.make_junctions(50000, 48)
    order = np.lexsort((strand, right, left, cr))
    for o in (order, order[::-1]):
        a = [np.ascontiguousarray(x[o]) for x in (cr, left, right, strand)]
        want = O.cluster_csr(*a)
        got = ctx.cluster(*a)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        assert np.array_equal(got[0], np.arange(50000) if o is order else np.arange(50000)[::-1])


def test_cluster_async_and_deferred_errors(ctx):
    from splicedice_amd.engine import SdiceError
    n = 20000
    cr, left, right, strand = synth.make_junctions(n, 49)
    want = O.cluster_csr(cr, left, right, strand)
    d = [ctx.to_device(x) for x in (cr, left, right, strand)]
    d_row_of, d_row_ptr = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_col, nnz = ctx.cluster_dev(*d, d_row_of, d_row_ptr, sync=False)
    assert nnz is None
    nnz, reach = ctx.cluster_status()
    assert nnz == want[2].size and reach > 0
    assert np.array_equal(d_row_of.to_host(), want[0]) and np.array_equal(d_row_ptr.to_host(), want[1])
    assert np.array_equal(d_col.offset(0, (nnz,)).to_host(), want[2])
    rows = np.repeat(np.arange(n), np.diff(want[1]))
    assert reach == int(np.abs(rows - want[2]).max())
    # a duplicate junction: synchronous call raises, asynchronous call reports at the next sync and
    # leaves an all-zero row_ptr behind
    cr2, left2, right2, strand2 = cr.copy(), left.copy(), right.copy(), strand.copy()
    cr2[7], left2[7], right2[7], strand2[7] = cr2[11], left2[11], right2[11], strand2[11]
    with pytest.raises(SdiceError, match="duplicate"):
        ctx.cluster(cr2, left2, right2, strand2)
    d2 = [ctx.to_device(x) for x in (cr2, left2, right2, strand2)]
    ctx.cluster_dev(*d2, d_row_of, d_row_ptr, sync=False)
    with pytest.raises(SdiceError, match="duplicate"):
        ctx.sync()
    assert not d_row_ptr.to_host().any()
    ctx.sync()                                          # the error was consumed
    # invalid coordinates, asynchronous
    left3 = left.copy()
    left3[5] = right[5] + 1
    ctx.cluster_dev(d[0], ctx.to_device(left3), d[2], d[3], d_row_of, d_row_ptr, sync=False)
    with pytest.raises(SdiceError, match="invalid junction"):
        ctx.cluster_status()
    # a list longer than the buffer: asynchronous call says so, a synchronous one sizes the buffer
    crd, ld, rd, sd = synth.make_junctions(3000, 50, n_chrom=1, gene_spacing=30, len_span=200000)
    wantd = O.cluster_csr(crd, ld, rd, sd)
    assert wantd[2].size > 16 * 3000 + 1024
    dd = [ctx.to_device(x) for x in (crd, ld, rd, sd)]
    d_row_of2, d_row_ptr2 = ctx.empty(3000, np.int32), ctx.empty(3001, np.int64)
    fresh = type(ctx)(0)                                 # a context whose list buffer has never grown
    try:
        fd = [fresh.to_device(x) for x in (crd, ld, rd, sd)]
        f_row_of, f_row_ptr = fresh.empty(3000, np.int32), fresh.empty(3001, np.int64)
        fresh.cluster_dev(*fd, f_row_of, f_row_ptr, sync=False)
        with pytest.raises(SdiceError, match="capacity"):
            fresh.sync()
        rp = f_row_ptr.to_host()
        assert (np.diff(rp) >= 0).all() and rp[-1] <= 16 * 3000 + 1024      # clamped, in bounds
        f_col, f_nnz = fresh.cluster_dev(*fd, f_row_of, f_row_ptr)           # synchronous: grows and re-runs
        assert f_nnz == wantd[2].size and np.array_equal(f_col.to_host(), wantd[2])
        assert np.array_equal(f_row_ptr.to_host(), wantd[1])
        fresh.cluster_dev(*fd, f_row_of, f_row_ptr, sync=False)               # now it fits
        assert fresh.cluster_status()[0] == wantd[2].size
    finally:
        fresh.close()


def test_cluster_async_error_survives_a_later_chain(ctx):
    """An asynchronous clustering that failed must still be reported after ANOTHER asynchronous clustering has been
    enqueued behind it (the later chain re-zeroes the per-chain status words: the failure lives in the sticky word)."""
    from splicedice_amd.engine import SdiceError
    n = 20000
    cr, left, right, strand = synth.make_junctions(n, 51)
    cr2, left2, right2, strand2 = cr.copy(), left.copy(), right.copy(), strand.copy()
    cr2[7], left2[7], right2[7], strand2[7] = cr2[11], left2[11], right2[11], strand2[11]
    good = [ctx.to_device(x) for x in (cr, left, right, strand)]
    bad = [ctx.to_device(x) for x in (cr2, left2, right2, strand2)]
    d_row_of, d_row_ptr = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_row_of2, d_row_ptr2 = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    ctx.sync()
    ctx.cluster_dev(*bad, d_row_of, d_row_ptr, sync=False)        # fails on the device
    ctx.cluster_dev(*good, d_row_of2, d_row_ptr2, sync=False)     # a valid chain behind it
    with pytest.raises(SdiceError, match="duplicate"):
        ctx.sync()
    ctx.sync()                                                    # consumed
    # ... and a SYNCHRONOUS call behind a failed asynchronous one reports it instead of dropping it
    ctx.cluster_dev(*bad, d_row_of, d_row_ptr, sync=False)
    with pytest.raises(SdiceError, match="duplicate"):
        ctx.cluster_dev(*good, d_row_of2, d_row_ptr2)
    want = O.cluster_csr(cr, left, right, strand)
    d_col, nnz = ctx.cluster_dev(*good, d_row_of2, d_row_ptr2)
    assert nnz == want[2].size and np.array_equal(d_row_ptr2.to_host(), want[1])


def test_cluster_lookback_tile_loop_and_give_up(ctx):
    """The neighbour kernel's workgroups walk their tiles in increasing order (a grid no larger than what the device
    holds at once: the look-back's forward progress): a grid of 3 / 8 workgroups over ~118 tiles gives the lists of the
    one-workgroup-per-tile launch.  And a look-back that gives up (forced: cluster.ablate = 256) is not silent: the
    synchronous call falls back to the generic chain and still returns the reference's lists, the asynchronous one
    reports it at the next sync."""
    from splicedice_amd.engine import SdiceError
    n = 60_000
    cr, left, right, strand = synth.make_junctions(n, 61)
    want = ctx.cluster(cr, left, right, strand)
    d = [ctx.to_device(x) for x in (cr, left, right, strand)]
    d_row_of, d_row_ptr = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    try:
        for grid in (3, 8):
            ctx.set_param("cluster.nb_grid", grid)
            got = ctx.cluster(cr, left, right, strand)
            assert all(np.array_equal(g, w) for g, w in zip(got, want)), grid
            d_row_ptr.zero()
            d_col, _ = ctx.cluster_dev(*d, d_row_of, d_row_ptr, sync=False)          # asynchronous chain, same grid
            nnz, _ = ctx.cluster_status()
            assert nnz == want[2].size and np.array_equal(d_row_ptr.to_host(), want[1])
            assert np.array_equal(d_col.offset(0, (nnz,)).to_host(), want[2])
        ctx.set_param("cluster.nb_grid", 0)
        ctx.set_param("cluster.ablate", 256)
        got = ctx.cluster(cr, left, right, strand)                                    # synchronous: generic path takes over
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
        ctx.cluster_dev(*d, d_row_of, d_row_ptr, sync=False)
        with pytest.raises(SdiceError, match="look-back"):
            ctx.sync()
    finally:
        ctx.set_param("cluster.nb_grid", 0)
        ctx.set_param("cluster.ablate", 0)
    got = ctx.cluster(cr, left, right, strand)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))


def test_cluster_list_size_is_bounded(ctx):
    """A clustering whose neighbour lists exceed the cap ends in a clean SDICE_ERR_NOMEM that names the size, before
    anything of that size is allocated on the device or the host (the reference degrades gracefully with Python lists,
    SPLICEDICE.py:230-255; an unbounded np.empty(nnz) / hipMalloc(4 nnz) does not)."""
    from splicedice_amd.engine import SdiceError
    crd, ld, rd, sd = synth.make_junctions(3000, 50, n_chrom=1, gene_spacing=30, len_span=200000)
    want = O.cluster_csr(crd, ld, rd, sd)
    assert want[2].size > 200_000
    fresh = type(ctx)(0)
    try:
        fresh.set_param("cluster.max_nnz", 100_000)
        with pytest.raises(SdiceError, match=r"neighbour list.*%d entries" % want[2].size):
            fresh.cluster(crd, ld, rd, sd)
        fresh.set_param("cluster.generic", 1)
        with pytest.raises(SdiceError, match=r"neighbour list.*%d entries" % want[2].size):
            fresh.cluster(crd, ld, rd, sd)
        fresh.set_param("cluster.generic", 0)
        fresh.set_param("cluster.max_nnz", 0)                     # automatic: what HBM and host memory hold
        got = fresh.cluster(crd, ld, rd, sd)
        assert np.array_equal(got[2], want[2])
    finally:
        fresh.close()


def test_cluster_empty_and_invalid(ctx):
    e32 = np.zeros(0, np.int32)
    row_of, row_ptr, col = ctx.cluster(e32, e32, e32, np.zeros(0, np.int8))
    assert row_of.size == 0 and row_ptr.tolist() == [0] and col.size == 0
    from splicedice_amd.engine import SdiceError
    with pytest.raises(SdiceError):
        ctx.cluster(np.zeros(2, np.int32), np.array([5, 1], np.int32), np.array([3, 2], np.int32), np.zeros(2, np.int8))


# ------------------------------------------------------------------------------ PS
@pytest.mark.parametrize("tag", ["a", "b", "c", "dense"])
def test_ps_golden(ctx, golden_dir, tag):
    z = _golden_arrays(golden_dir, tag)
    ps, excl = ctx.ps(z["counts_rows"], z["row_ptr"], z["col"], want_excl=True)
    assert ps.tobytes() == z["psi"].tobytes()          # bit-exact incl. NaN payload-insensitive check below
    _, want_excl = O.calculate_psi_vectorised(z["counts_rows"], z["row_ptr"], z["col"])
    assert np.array_equal(excl, want_excl)


@pytest.mark.parametrize("n,s,seed", [(3000, 100, 1), (5000, 4, 2), (777, 7, 3), (300, 1000, 4), (2000, 1, 5),
                                      (1500, 260, 6), (129, 36, 7), (40000, 16, 8)])
def test_ps_vs_oracle(ctx, n, s, seed):
    cr, left, right, strand = synth.make_junctions(n, seed, n_chrom=4)
    _, row_ptr, col = O.cluster_csr(cr, left, right, strand) if n <= 5000 else ctx.cluster(cr, left, right, strand)
    counts = synth.make_counts(n, s, seed + 50)
    want_ps, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
    assert np.array_equal(excl, want_excl)
    assert ps.view(np.uint32).tolist() == want_ps.view(np.uint32).tolist() or \
        np.array_equal(np.isnan(ps), np.isnan(want_ps)) and np.array_equal(ps[~np.isnan(ps)], want_ps[~np.isnan(ps)])
    only_ps = ctx.ps(counts, row_ptr, col)
    assert np.array_equal(only_ps, ps, equal_nan=True)
    only_excl = ctx.ps(counts, row_ptr, col, want_excl=True, want_ps=False)
    assert np.array_equal(only_excl, excl)


@pytest.mark.parametrize("n,s", [(30000, 100), (20000, 500), (50000, 36), (9000, 1000), (40000, 8)])
def test_ps_kernel_generations_agree(ctx, n, s):
    """the two tile kernels behind sdice_ps_dev -- second-generation register-staged (the default) and first generation
    (ps.gen1 = 1: any row width, any tile shape) -- on the device path WITH the clustering's reach words (halo sized per
    tile), without them (ps.use_reach = 0), with exclusion sums, and with the fused '.3f' store: bit-identical"""
    cr, left, right, strand = synth.make_junctions(n, n + s)
    d = [ctx.to_device(x) for x in (cr, left, right, strand)]
    d_row_of, d_rp = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_col, nnz = ctx.cluster_dev(*d, d_row_of, d_rp)
    counts = synth.make_counts(n, s, n + s + 1)
    counts[7, :] = (1 << 24) - 1                       # one tile whose bound forces the 64-bit path
    d_counts = ctx.to_device(counts)
    d_ps, d_excl = ctx.empty((n, s), np.float32), ctx.empty((n, s), np.int64)
    refs = {}
    try:
        for kern, reach, q3 in ((1, 1, 0), (0, 1, 0), (0, 0, 0), (1, 1, 1), (0, 1, 1), (0, 0, 1)):
            ctx.set_param("ps.gen1", kern)
            ctx.set_param("ps.use_reach", reach)
            ctx.set_param("ps.quantize3", q3)
            d_ps.memset(0xff)
            d_excl.memset(0xff)
            ctx.ps_dev(d_counts, d_rp, d_col, d_excl, d_ps)
            ctx.sync()
            got = (d_ps.to_host().view(np.uint32), d_excl.to_host())
            if q3 not in refs:
                refs[q3] = got
            assert np.array_equal(got[0], refs[q3][0]) and np.array_equal(got[1], refs[q3][1]), (kern, reach, q3)
    finally:
        ctx.set_param("ps.gen1", 0)
        ctx.set_param("ps.use_reach", 1)
        ctx.set_param("ps.quantize3", 0)
    ref = refs[0]
    row_ptr, col = d_rp.to_host(), d_col.to_host()
    want_ps, want_excl = O.calculate_psi_vectorised(counts[:3000], row_ptr[:3001], np.minimum(col[: int(row_ptr[3000])], 2999))
    inside = np.array([(col[row_ptr[r]:row_ptr[r + 1]] < 3000).all() for r in range(3000)])
    assert np.array_equal(ref[1][:3000][inside], want_excl[inside])


@pytest.mark.parametrize("lds,threads,tile_rows,halo,s", [
    (8192, 64, 0, -1, 100), (32768, 256, 7, 0, 100), (65536, 1024, 0, 3, 100), (163840, 512, 0, 40, 100),
    (81920, 1024, 0, 1, 100),
    # tile shapes at the limits of the second-generation kernel's register staging (four own-row vectors, two row
    # pointers per thread): few threads x wide rows, many rows x narrow rows, a forced tile beyond both
    (81920, 64, 0, -1, 256), (81920, 64, 0, -1, 132), (163840, 128, 0, -1, 200), (81920, 256, 0, -1, 4),
    (81920, 256, 0, -1, 8), (81920, 64, 0, 16, 8), (81920, 128, 300, -1, 8), (81920, 64, 40, -1, 256),
    (163840, 1024, 0, -1, 1000), (81920, 192, 0, -1, 520)])
def test_ps_launch_shapes(ctx, lds, threads, tile_rows, halo, s):
    n = 4000
    cr, left, right, strand = synth.make_junctions(n, 21, n_chrom=3)
    _, row_ptr, col = O.cluster_csr(cr, left, right, strand)
    counts = synth.make_counts(n, s, 22)
    want_ps, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    try:
        ctx.set_param("ps.lds_bytes", lds)
        ctx.set_param("ps.threads", threads)
        ctx.set_param("ps.tile_rows", tile_rows)
        ctx.set_param("ps.halo_rows", halo)
        ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
    finally:
        ctx.set_param("ps.lds_bytes", 81920)
        ctx.set_param("ps.threads", 1024)
        ctx.set_param("ps.tile_rows", 0)
        ctx.set_param("ps.halo_rows", -1)
    assert np.array_equal(excl, want_excl)
    assert np.array_equal(ps, want_ps, equal_nan=True)


def test_ps_arbitrary_csr(ctx):
    """Any valid CSR is accepted (e.g. parsed from a user's _allClusters.tsv): neighbours far
    outside the tile window take the global-memory path; long lists overflow the LDS col stage."""
    rng = np.random.default_rng(5)
    n, s = 3000, 20
    deg = rng.integers(0, 6, size=n)
    deg[17] = 2500           # one huge list
    deg[18] = 0
    row_ptr = np.r_[0, np.cumsum(deg)].astype(np.int64)
    col = rng.integers(0, n, size=int(row_ptr[-1])).astype(np.int32)
    counts = synth.make_counts(n, s, 6)
    want_ps, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
    assert np.array_equal(excl, want_excl)
    assert np.array_equal(ps, want_ps, equal_nan=True)


def test_ps_big_counts_and_zero(ctx):
    # counts up to 2^24-1 with a long list: sums exceed 2^32; all-zero cluster -> NaN
    n, s = 600, 8
    counts = np.full((n, s), (1 << 24) - 1, np.int32)
    counts[-3:] = 0
    row_ptr = np.zeros(n + 1, np.int64)
    row_ptr[1:] = 500
    row_ptr[1:] = np.cumsum(np.r_[500, np.zeros(n - 4, np.int64), 2, 2, 2])
    col = np.r_[np.arange(1, 501), [n - 2, n - 1], [n - 3, n - 1], [n - 3, n - 2]].astype(np.int32)
    want_ps, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
    assert excl[0, 0] == 500 * ((1 << 24) - 1) and np.array_equal(excl, want_excl)
    assert np.isnan(ps[-1]).all() and np.array_equal(ps, want_ps, equal_nan=True)


@pytest.mark.parametrize("n,s,big", [(3000, 100, False), (1500, 300, False), (400, 12, True)])
def test_ps_fused_quantise(ctx, n, s, big):
    """param ps.quantize3: the PS store writes float32(f'{ps:.3f}') directly (K4 fused into K3);
    must equal PS followed by the separate quantise pass, on the 32-bit and the 64-bit sum paths."""
    cr, left, right, strand = synth.make_junctions(n, 61, n_chrom=3)
    _, row_ptr, col = O.cluster_csr(cr, left, right, strand)
    counts = synth.make_counts(n, s, 62)
    if big:
        counts = counts * 100000 + 7          # sums beyond 2^24: the float64 path
    want = O.quantize3_fast(O.calculate_psi_vectorised(counts, row_ptr, col)[0])
    two_pass = ctx.quantize3(ctx.ps(counts, row_ptr, col))
    ctx.set_param("ps.quantize3", 1)
    try:
        fused = ctx.ps(counts, row_ptr, col)
    finally:
        ctx.set_param("ps.quantize3", 0)
    assert np.array_equal(two_pass, want, equal_nan=True)
    assert np.array_equal(fused, want, equal_nan=True)


def test_ps_fused_quantise_on_rounding_boundaries(ctx):
    """the fused '.3f' round trip works in float32 (exact product by fma, half integers decided by the residual): PS values
    k/d that sit exactly on a rounding boundary (d = 16, 32, ...: x * 1000 is a half integer), next to one (d = 2000, 4000,
    ...: float32(a/d) * 1000 rounds to a half integer in float32 but the exact product lies beside it) and everything in
    between, against the float64 restatement and the separate quantise kernel"""
    S = 128
    dens = [16, 32, 64, 80, 160, 625, 1600, 2000, 3125, 4000, 6000, 14000, 16000]
    pairs = []                                             # (a, b): PS = a / (a + b) for one row, b / (a + b) for its partner
    for d in dens:
        pairs += [(a, d - a) for a in range(d + 1)]
    n_pairs = -(-len(pairs) // S)
    pairs += [(1, 1)] * (n_pairs * S - len(pairs))
    ab = np.array(pairs, dtype=np.int32).reshape(n_pairs, S, 2)
    counts = np.empty((2 * n_pairs, S), np.int32)
    counts[0::2] = ab[:, :, 0]
    counts[1::2] = ab[:, :, 1]
    n = 2 * n_pairs
    row_ptr = np.arange(n + 1, dtype=np.int64)
    col = (np.arange(n, dtype=np.int32) ^ 1)               # rows 2i and 2i + 1 are each other's only neighbour
    psi = O.calculate_psi_vectorised(counts, row_ptr, col)[0]
    want = O.quantize3_fast(psi)
    x1000 = psi[np.isfinite(psi)].astype(np.float64) * 1000.0
    assert (np.abs(x1000 - np.floor(x1000) - 0.5) == 0).sum() > 50            # exact ties are in the fixture
    two_pass = ctx.quantize3(ctx.ps(counts, row_ptr, col))
    ctx.set_param("ps.quantize3", 1)
    try:
        fused = ctx.ps(counts, row_ptr, col)
    finally:
        ctx.set_param("ps.quantize3", 0)
    assert np.array_equal(two_pass, want, equal_nan=True)
    assert np.array_equal(fused, want, equal_nan=True)


def test_ps_empty(ctx):
    ps = ctx.ps(np.zeros((0, 5), np.int32), np.zeros(1, np.int64), np.zeros(0, np.int32))
    assert ps.shape == (0, 5)


@pytest.mark.parametrize("n,s,seed", [(900, 5, 1), (400, 100, 2), (300, 333, 3), (1200, 1, 4), (64, 64, 5)])
def test_ps_f64_fractional_counts_vs_oracle(ctx, n, s, seed):
    """counts_to_ps.py:58-70 on a float64 table: the sums round, so the ORDER of the additions shows in the last
    bit -- bit-exact against the restated loop, on values spanning 16 decades, with negative zero, NaN and inf
    cells, lists of 0..40 entries (the kernel adds them four at a time), source-only rows past n_out."""
    rng = np.random.default_rng(seed)
    n_out = n - n // 7
    deg = rng.integers(0, 9, size=n_out)
    deg[rng.integers(0, n_out, size=5)] = rng.integers(20, 41, size=5)
    row_ptr = np.r_[0, np.cumsum(deg)].astype(np.int64)
    col = rng.integers(0, n, size=int(row_ptr[-1])).astype(np.int32)
    counts = rng.gamma(0.7, 30.0, size=(n, s)) * 10.0 ** rng.integers(-8, 9, size=(n, 1))
    counts[rng.random((n, s)) < 0.2] = 0.0
    flat = counts.reshape(-1)
    flat[rng.integers(0, flat.size, size=6)] = [np.nan, np.inf, -0.0, -3.25, 1e-310, 1.7e308]
    want = O.write_ps_values_f64(counts, row_ptr, col, n_out)
    got = ctx.ps_f64(counts, row_ptr, col, n_out=n_out)
    assert got.shape == (n_out, s)
    assert got.view(np.uint64).tolist() == want.view(np.uint64).tolist() or \
        (np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]))


def test_ps_f64_integer_table_equals_integer_kernel(ctx):
    """On an integer-valued table the float64 sums are exact: same numbers as ps_tile_kernel's int64 sums."""
    n, s = 3000, 37
    cr, left, right, strand = synth.make_junctions(n, 9, n_chrom=3)
    _, row_ptr, col = O.cluster_csr(cr, left, right, strand)
    counts = synth.make_counts(n, s, 10)
    excl = ctx.ps(counts, row_ptr, col, want_excl=True, want_ps=False)
    own = counts.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        want = own / (own + excl.astype(np.float64))
    got = ctx.ps_f64(own, row_ptr, col)
    assert np.array_equal(got, want, equal_nan=True)


def test_ps_f64_empty_and_bad_csr(ctx):
    assert ctx.ps_f64(np.zeros((0, 4)), np.zeros(1, np.int64), np.zeros(0, np.int32)).shape == (0, 4)
    assert ctx.ps_f64(np.ones((3, 0)), np.zeros(4, np.int64), np.zeros(0, np.int32)).shape == (3, 0)
    with pytest.raises(Exception, match="out of range"):
        ctx.ps_f64(np.ones((3, 2)), np.array([0, 1, 1, 1], np.int64), np.array([3], np.int32))
    with pytest.raises(Exception, match="non-decreasing"):
        ctx.ps_f64(np.ones((3, 2)), np.array([0, 2, 1, 2], np.int64), np.array([0, 1], np.int32))


def test_mark_low_and_quantize(ctx):
    rng = np.random.default_rng(3)
    x = rng.random((50, 9)).astype(np.float32)
    idx = np.array([0, 5, 449, 77], np.int64)
    y = ctx.mark_low(x.copy(), idx)
    assert np.isnan(y.ravel()[idx]).all() and np.isnan(y).sum() == 4
    # '.3f' text round trip: every k/1000 neighbourhood + random values + specials
    k = np.arange(0, 1001, dtype=np.float64) / 1000.0
    base = k.astype(np.float32)
    vals = np.concatenate([base, np.nextafter(base, np.float32(2)), np.nextafter(base, np.float32(-1)),
                           (k + 0.0005).astype(np.float32), rng.random(20000).astype(np.float32),
                           np.float32([np.nan, 0.0, 1.0, 0.0005, 0.9995, 0.99951, 1e-8])])
    got = ctx.quantize3(vals)
    want = O.quantize3(vals)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(O.quantize3_fast(vals), want, equal_nan=True)


# ------------------------------------------------------------------------------ rank-sum
def _check_ranksum(got, want):
    assert np.array_equal(got["tested"], want["tested"])
    t = want["tested"].astype(bool)
    for k in ("med1", "med2", "mean1", "mean2", "delta"):
        assert np.array_equal(got[k][t], want[k][t]), k       # float32, bit-exact
        assert not got[k][~t].any()
    assert np.array_equal(got["z"][t], want["z"][t])          # z bit-exact (exact numerator, IEEE sqrt/div)
    np.testing.assert_allclose(got["p"][t], want["p"][t], rtol=P_RTOL_TIGHT, atol=0)
    assert P_RTOL_TIGHT <= P_RTOL


def test_ranksum_kat(ctx, golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "kat_ranksums.json"))):
        x, y = np.float32(c["x"]), np.float32(c["y"])
        row = np.concatenate([x, y])[None, :]
        for variant in ((0, 2, 3, 4) if max(len(x), len(y)) <= 64 else (2, 3)):
            ctx.set_param("ranksum.variant", variant)
            try:
                r = ctx.ranksum(row, np.arange(len(x)), np.arange(len(x), len(x) + len(y)))
            finally:
                ctx.set_param("ranksum.variant", 0)
            assert r["tested"][0] == 1
            assert r["z"][0] == c["z"]
            assert abs(r["p"][0] - c["p"]) <= P_RTOL_TIGHT * c["p"]
            assert r["med1"][0] == np.float32(c["med1"]) and r["med2"][0] == np.float32(c["med2"])
            assert r["mean1"][0] == np.float32(c["mean1"]) and r["mean2"][0] == np.float32(c["mean2"])


@pytest.mark.parametrize("n1,n2,s,variant", [(50, 50, 100, 0), (50, 50, 100, 2), (3, 3, 6, 0), (3, 3, 6, 2),
                                              (64, 64, 130, 0), (63, 64, 130, 0), (64, 63, 130, 0), (7, 33, 64, 0), (9, 17, 40, 0), (65, 10, 80, 0),
                                              (500, 500, 1000, 0), (500, 500, 1000, 2), (200, 130, 400, 0),
                                              (200, 130, 400, 2), (1500, 3, 1600, 0), (1024, 1000, 2100, 0),
                                              (50, 50, 100, 3), (3, 3, 6, 3), (128, 65, 200, 3), (300, 7, 400, 3),
                                              (50, 50, 100, 4), (3, 3, 6, 4), (64, 64, 130, 4), (7, 33, 64, 4),
                                              (9, 17, 40, 4), (20, 20, 40, 4)])
def test_ranksum_vs_oracle(ctx, n1, n2, s, variant):
    n = 700 if s <= 200 else 60
    ps = synth.make_ps_matrix(n, s, seed=n1 * 1000 + n2, nan_frac=0.1)
    ps[0, :] = 0.5                 # all ties -> z = 0, p = 1
    ps[1, :] = np.nan              # untested
    ps[2, : s // 2] = np.nan
    rng = np.random.default_rng(n1 + n2)
    cols = rng.permutation(s)
    g1 = np.sort(cols[:n1])        # table order (compareSampleSets.py:96-102)
    g2 = np.sort(cols[n1:n1 + n2])
    want = O.compare_rows(ps, g1, g2)
    ctx.set_param("ranksum.variant", variant)
    try:
        got = ctx.ranksum(ps, g1, g2)
    finally:
        ctx.set_param("ranksum.variant", 0)
    _check_ranksum(got, want)


def test_ranksum_small_groups_untested(ctx):
    ps = synth.make_ps_matrix(10, 8, seed=1)
    got = ctx.ranksum(ps, [0, 1], [2, 3, 4, 5])
    assert not got["tested"].any() and not got["p"].any()


# ------------------------------------------------------------------------------ Fisher
def test_fisher_kat(ctx, golden_dir):
    kats = json.load(open(os.path.join(golden_dir, "kat_fisher.json")))
    tables = np.array([t for t, _ in kats], np.int64)
    want = np.array([p for _, p in kats])
    got = ctx.fisher_tables(tables)
    np.testing.assert_allclose(got, want, rtol=P_RTOL_TIGHT, atol=0)
    ctx.set_param("fisher.table_max", 64)      # force the device-lgamma path for large margins
    try:
        got2 = ctx.fisher_tables(tables)
    finally:
        ctx.set_param("fisher.table_max", 1 << 20)
    np.testing.assert_allclose(got2, want, rtol=P_RTOL_TIGHT, atol=0)


@pytest.mark.parametrize("n,s,mean", [(40, 6, 20), (6, 31, 5), (3, 2, 100), (5, 12, 400)])
def test_fisher_pairs_vs_scipy(ctx, n, s, mean):
    incl = synth.make_counts(n, s, 77 + s, mean=mean)
    excl = synth.make_counts(n, s, 78 + s, mean=mean * 4).astype(np.int64)
    excl[0, :] = 0
    want = O.fisher_pairs(incl, excl)
    got = ctx.fisher_pairs(incl, excl)
    np.testing.assert_allclose(got, want, rtol=P_RTOL_TIGHT, atol=0)
    # a log-factorial table of 64 entries: tables with a larger total leave the pair kernel as markers and are finished
    # by fisher_beyond_table_kernel (rows with such tables only; here some, all or -- margins of zero -- none of a row's pairs)
    ctx.set_param("fisher.table_max", 64)
    try:
        got2 = ctx.fisher_pairs(incl, excl)
    finally:
        ctx.set_param("fisher.table_max", 1 << 20)
    assert (got2 >= 0).all()
    np.testing.assert_allclose(got2, want, rtol=P_RTOL_TIGHT, atol=0)


def test_fisher_pairs_walks_across_the_mode_with_tiny_p(ctx):
    """pairs whose observed table lies hundreds of standard deviations from the mode: the up-walk crosses the mode with
    the numerator product outgrowing the denominator by more than 2^500 (the exponent count of the walk, the scaling
    between the halves of a trip), p-values down to ~1e-267; every lane of the wave busy with a different mix"""
    rng = np.random.default_rng(4242)
    s = 12
    incl = np.empty((6, s), np.int32)
    excl = np.empty((6, s), np.int64)
    for r in range(6):
        hi = rng.random(s) < 0.5
        incl[r] = np.where(hi, rng.integers(300, 460, s), rng.integers(0, 12, s))
        excl[r] = np.where(hi, rng.integers(0, 12, s), rng.integers(300, 460, s))
    want = O.fisher_pairs(incl, excl)
    got = ctx.fisher_pairs(incl, excl)
    assert (want < 1e-250).any() and (want > 1e-3).any() and (want >= 1e-280).all()   # (scipy's own range: DESIGN.md section 7)
    np.testing.assert_allclose(got, want, rtol=P_RTOL_TIGHT, atol=0)


def test_fisher_step_counts(ctx):
    """fisher.count_steps: the pair kernel's own count of issued and useful lane-steps (bench.py's useful_lane_frac) -- the
    useful ones can be no more than the support sizes allow and no fewer than one trip's worth per walked side; results
    are the same numbers with the counting build"""
    incl = synth.make_counts(50, 24, 5, mean=40)
    excl = synth.make_counts(50, 24, 6, mean=160).astype(np.int64)
    plain = ctx.fisher_pairs(incl, excl)
    ctx.set_param("fisher.count_steps", 1)
    try:
        counted = ctx.fisher_pairs(incl, excl)
        useful, issued = ctx.fisher_step_stats()
    finally:
        ctx.set_param("fisher.count_steps", 0)
    assert np.array_equal(plain, counted)
    a = incl[:, :, None].astype(np.int64); b = incl[:, None, :].astype(np.int64)
    c = excl[:, :, None]; d = excl[:, None, :]
    iu = np.triu_indices(24, 1)
    support = (np.minimum(a, d) + np.minimum(b, c))[:, iu[0], iu[1]].sum()         # steps if no tail were cut
    assert 0 < useful <= support and useful <= issued and issued % 64 == 0


def test_fisher_pairs_long_walks_and_sparse_rows(ctx):
    """the pair kernel's state machine on walks of hundreds of steps (counts in the thousands: the products P, Q, S are
    rescaled every few steps, the negligible-tail cut ends the walks) and on sparse rows (most pairs have a zero margin
    and never enter a walk; one-sided walks when a sits on the edge of the support)"""
    rng = np.random.default_rng(99)
    s = 9
    incl = rng.poisson(rng.choice([2500.0, 3000.0, 3300.0], size=(3, s))).astype(np.int32)
    excl = rng.poisson(rng.choice([9000.0, 12000.0], size=(3, s))).astype(np.int64)
    sparse_i = (rng.random((40, s)) < 0.25) * rng.integers(1, 30, size=(40, s))
    sparse_e = (rng.random((40, s)) < 0.5) * rng.integers(1, 60, size=(40, s))
    incl = np.vstack([incl, sparse_i.astype(np.int32)])
    excl = np.vstack([excl, sparse_e.astype(np.int64)])
    want = O.fisher_pairs(incl, excl)
    got = ctx.fisher_pairs(incl, excl)
    assert (want[3:] == 1.0).mean() > 0.3 and (want[:3] < 1e-3).any()        # the fixture exercises both
    np.testing.assert_allclose(got, want, rtol=P_RTOL_TIGHT, atol=0)


# ------------------------------------------------------------------------------ BH
@pytest.mark.parametrize("m", [1, 2, 255, 256, 257, 5000, 100000])
def test_bh_vs_oracle(ctx, m):
    rng = np.random.default_rng(m)
    p = rng.random(m) ** 3
    p[rng.random(m) < 0.1] = 1.0
    if m > 10:
        p[:5] = p[5]               # ties
        p[7] = 0.0
    want = O.bh_fdr(p)
    got = ctx.bh(p)
    np.testing.assert_allclose(got, want, rtol=1e-14, atol=0)
    from scipy.stats import false_discovery_control
    np.testing.assert_allclose(got, false_discovery_control(p, method="bh"), rtol=1e-12, atol=0)


def test_bh_columns(ctx):
    rng = np.random.default_rng(9)
    p = rng.random((300, 15)) ** 2
    np.testing.assert_allclose(ctx.bh_columns(p), O.bh_columns(p), rtol=1e-14, atol=0)


# ------------------------------------------------------------------------------ device-resident pipeline
def test_device_pipeline_matches_host_calls(ctx):
    n, s = 6000, 40
    cr, left, right, strand = synth.make_junctions(n, 91, n_chrom=5)
    counts_in = synth.make_counts(n, s, 92)
    row_of, row_ptr, col = ctx.cluster(cr, left, right, strand)
    counts_rows = np.zeros_like(counts_in)
    counts_rows[row_of] = counts_in
    ps_host = ctx.ps(counts_rows, row_ptr, col)
    d = {k: ctx.to_device(v) for k, v in dict(c=cr, l=left, r=right, s=strand).items()}
    d_row_of, d_row_ptr = ctx.empty(n, np.int32), ctx.empty(n + 1, np.int64)
    d_col, nnz = ctx.cluster_dev(d["c"], d["l"], d["r"], d["s"], d_row_of, d_row_ptr)
    assert nnz == col.size and np.array_equal(d_col.to_host(), col)
    d_counts, d_ps = ctx.to_device(counts_rows), ctx.empty((n, s), np.float32)
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.ps_dev(d_counts, d_row_ptr, d_col, None, d_ps)
    ctx.quantize3_dev(d_ps)
    ctx.sync()
    rep = ctx.prof_report()
    ctx.prof_enable(False)
    assert rep["ps_tile_v3_kernel"][0] == 1 and rep["ps_tile_v3_kernel"][1] > 0
    assert np.array_equal(d_ps.to_host(), O.quantize3_fast(ps_host), equal_nan=True)
    g1, g2 = np.arange(0, 20, dtype=np.int32), np.arange(20, 40, dtype=np.int32)
    out = dict(tested=ctx.empty(n, np.uint8), p=ctx.empty(n, np.float64), z=ctx.empty(n, np.float64),
               med1=ctx.empty(n, np.float32), med2=ctx.empty(n, np.float32), mean1=ctx.empty(n, np.float32),
               mean2=ctx.empty(n, np.float32), delta=ctx.empty(n, np.float32))
    ctx.ranksum_dev(d_ps, ctx.to_device(g1), ctx.to_device(g2), out)
    want = O.compare_rows(O.quantize3_fast(ps_host), g1, g2)
    got = {k: v.to_host() for k, v in out.items()}
    _check_ranksum(got, want)
    # world-size-1 all-gather is a device copy
    d_recv = ctx.empty(n, np.float64)
    ctx.allgather_dev(out["p"], d_recv)
    assert np.array_equal(d_recv.to_host(), got["p"])


# ------------------------------------------------------------------------------ chi2 (pairwise --chi2)
def test_chi2_pairs_vs_scipy(ctx):
    incl = synth.make_counts(30, 9, 501, mean=25) + 1
    excl = (synth.make_counts(30, 9, 502, mean=90) + 1).astype(np.int64)
    p, n_bad = ctx.chi2_pairs(incl, excl)
    assert n_bad == 0
    np.testing.assert_allclose(p, O.chi2_pairs(incl, excl), rtol=P_RTOL_TIGHT, atol=0)
    excl[4, :] = 0
    incl[4, 2] = 0
    p, n_bad = ctx.chi2_pairs(incl, excl)
    assert n_bad == 36 and np.isnan(p[4]).all() and not np.isnan(np.delete(p, 4, axis=0)).any()


def test_ps_fast_path_division_is_exact(ctx):
    """The float32 fast path (no v_div_scale / v_div_fixup) must equal float32(float64 quotient)
    over its whole domain: numerators and denominators up to 2^24, incl. 0/0 and 0/x."""
    rng = np.random.default_rng(11)
    n, s = 4096, 64
    # ring of junctions: row r has neighbours r-1 and r+1 -> degree 2, bound = (2^24-1)/3
    row_ptr = np.arange(0, 2 * n + 1, 2, dtype=np.int64)
    col = np.stack([(np.arange(n) - 1) % n, (np.arange(n) + 1) % n], axis=1).ravel().astype(np.int32)
    top = (1 << 24) // 3 - 1
    counts = rng.integers(0, top, size=(n, s), dtype=np.int32)
    counts[rng.random((n, s)) < 0.3] = 0
    counts[:64] = rng.integers(0, 4, size=(64, s))            # small integers: many exact ties and 0/0
    counts[100:164, :] = top
    want_ps, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
    assert np.array_equal(excl, want_excl)
    assert np.array_equal(ps, want_ps, equal_nan=True)
    # every quotient a/(a+b) with small a, b, through the same path
    a, b = np.meshgrid(np.arange(0, 257), np.arange(0, 257))
    c2 = np.zeros((2 * a.size, 4), np.int32)
    c2[0::2, :] = a.ravel()[:, None]
    c2[1::2, :] = b.ravel()[:, None]
    rp = np.zeros(2 * a.size + 1, np.int64)
    rp[1:] = np.cumsum(np.tile([1, 0], a.size))
    cl = np.arange(1, 2 * a.size, 2, dtype=np.int32)           # row 2i has the single neighbour 2i+1
    ps2 = ctx.ps(c2, rp, cl)
    with np.errstate(invalid="ignore", divide="ignore"):
        w = (a.ravel().astype(np.float64) / (a.ravel() + b.ravel()).astype(np.float64)).astype(np.float32)
    assert np.array_equal(ps2[0::2, 0], w, equal_nan=True)


@pytest.mark.parametrize("n,cols", [(3000, 190), (1, 5), (7000, 3), (50, 1000)])
def test_bh_columns_batched(ctx, n, cols):
    rng = np.random.default_rng(n + cols)
    p = rng.random((n, cols)) ** 2
    p[rng.random((n, cols)) < 0.05] = 1.0
    if n > 10:
        p[:4, 0] = p[4, 0]
    np.testing.assert_allclose(ctx.bh_columns(p), O.bh_columns(p), rtol=1e-14, atol=0)
    d = ctx.to_device(p)
    ctx.bh_columns_dev(d)
    np.testing.assert_allclose(d.to_host(), O.bh_columns(p), rtol=1e-14, atol=0)


@pytest.mark.parametrize("n,cols", [(1, 3), (64, 5), (257, 40), (1024, 9), (1025, 9), (5000, 33), (25000, 12), (70001, 3),
                                    (200000, 2)])
def test_bh_columns_samplesort_vs_generic(ctx, n, cols):
    """the sample-sort column path (bh_cols.hip) against the generic radix path, bit for bit, and the oracle:
    continuous values, heavy ties (p = 1, a few discrete levels as Fisher gives), one-value and two-value
    columns, values a few ulps apart, NaN, crowds of distinct values far narrower than a bin (the second-level sort)"""
    rng = np.random.default_rng(n * 31 + cols)
    p = rng.random((n, cols)) ** 2
    p[rng.random((n, cols)) < 0.3] = 1.0
    if cols > 1:
        p[:, 1] = rng.choice([1.0, 0.5, 0.0286, 0.2, 1e-5], size=n)        # discrete levels
    if cols > 2:
        p[:, 2] = 0.25                                                       # one value
    if cols > 3:
        p[:, 3] = 0.5 + rng.integers(0, 7, size=n) * 2.0 ** -53              # a few ulps apart
    if cols > 4 and n > 10:
        p[rng.integers(0, n, size=3), 4] = np.nan
    if cols > 5:
        # what a column of Fisher p-values looks like at its top: exact ones, a crowd of DISTINCT sums within 1e-13 of 1
        # (many keys in ONE bin of the bucket's counting sort, next to a tie group), ordinary values below
        q = rng.random(n)
        p[q < 0.06, 5] = 1.0
        crowd = (q >= 0.06) & (q < 0.14)
        p[crowd, 5] = 1.0 - rng.integers(1, 900, size=int(crowd.sum())) * 2.0 ** -53
    if cols > 6:
        # a crowd with a long tail around an ordinary value, and a stray neighbour inside the same bin
        crowd = rng.random(n) < 0.3
        p[crowd, 6] = 0.3 + (rng.standard_cauchy(int(crowd.sum())) * 40).astype(np.int64).clip(-10 ** 7, 10 ** 7) * 2.0 ** -54
    try:
        ctx.set_param("bh.columns_path", 1)
        d = ctx.to_device(p)
        ctx.bh_columns_dev(d)
        generic = d.to_host()
        ctx.set_param("bh.columns_path", 2)
        d = ctx.to_device(p)
        ctx.bh_columns_dev(d)
        fast = d.to_host()
        assert np.array_equal(generic, fast, equal_nan=True)
        for wg, mean in ((256, 100), (256, 900), (512, 0), (1024, 0), (256, 3000)):     # threads of a bucket workgroup (4 values
            ctx.set_param("bh.wg", wg)                                        # each), mean bucket (0: half its capacity); 900 and
            ctx.set_param("bh.mean", mean)                                    # 3000 fill the second kernel's list
            d = ctx.to_device(p)
            ctx.bh_columns_dev(d)
            assert np.array_equal(generic, d.to_host(), equal_nan=True), (wg, mean)
        ctx.set_param("bh.wg", 256)
        ctx.set_param("bh.mean", 900)
        ctx.set_param("bh.big_wg", 256)                                       # the listed buckets at 256 threads x 8 values
        d = ctx.to_device(p)
        ctx.bh_columns_dev(d)
        assert np.array_equal(generic, d.to_host(), equal_nan=True)
        ctx.set_param("bh.big_wg", 512)
        ctx.set_param("bh.mean", 0)
        ctx.set_param("bh.fused_count", 0)                                    # transpose and count as two kernels
        d = ctx.to_device(p)
        ctx.bh_columns_dev(d)
        assert np.array_equal(generic, d.to_host(), equal_nan=True)
        ctx.set_param("bh.fused_count", 1)
        if n > 2000:
            ctx.set_param("bh.reg_cap", 64)                                   # most buckets through the in-HBM path
            d = ctx.to_device(p)
            ctx.bh_columns_dev(d)
            assert np.array_equal(generic, d.to_host(), equal_nan=True)
    finally:
        ctx.set_param("bh.columns_path", 0)
        ctx.set_param("bh.reg_cap", 2048)
        ctx.set_param("bh.mean", 0)
        ctx.set_param("bh.wg", 256)
        ctx.set_param("bh.big_wg", 512)
        ctx.set_param("bh.fused_count", 1)
    ok = ~np.isnan(p).any(axis=0)
    np.testing.assert_allclose(fast[:, ok], O.bh_columns(p[:, ok]), rtol=1e-14, atol=0)


# ------------------------------------------------------------------------------ junction union
@pytest.mark.parametrize("n,distinct", [(1, 1), (2, 1), (5000, 40), (300_000, 90_000), (1_000_000, 1_000_000)])
def test_sort_unique_u64(ctx, n, distinct):
    rng = np.random.default_rng(n + distinct)
    pool = rng.integers(0, 1 << 63, size=distinct, dtype=np.uint64) | (rng.integers(0, 2, size=distinct, dtype=np.uint64) << np.uint64(63))
    keys = pool[rng.integers(0, distinct, size=n)]
    got = ctx.sort_unique_u64(keys)
    assert np.array_equal(got, np.unique(keys))
    assert ctx.sort_unique_u64(np.zeros(0, np.uint64)).size == 0
    same = ctx.sort_unique_u64(np.full(777, 12345, np.uint64))
    assert same.tolist() == [12345]


def test_ingest_union_gpu_equals_host(ctx, golden_dir):
    """juncio.ingest with the engine (packed keys, GPU sort + unique) == the numpy path."""
    from splicedice_amd import juncio, quant
    import argparse
    qdir = os.path.join(golden_dir, "quant_c1")
    p = argparse.ArgumentParser()
    quant.add_parser(p)
    args = p.parse_args(["-m", os.path.join(qdir, "manifest.rel.tsv"), "-o", "x"])
    cwd = os.getcwd()
    os.chdir(os.path.join(qdir, "inputs"))
    try:
        manifest = quant.parse_manifest(args.manifest)
        a = juncio.ingest(manifest, args, ctx)
        b = juncio.ingest(manifest, args, None)
    finally:
        os.chdir(cwd)
    assert a[0] == b[0] and a[1][0].size > 500
    for x, y in zip(a[1], b[1]):
        assert x.dtype == y.dtype and np.array_equal(x, y)


def test_ranksum_fuzz_shapes(ctx):
    """random group sizes across the kernel boundaries (pair <= 64 < wave <= 1024 < block), random NaN
    density, ties from 3-decimal quantisation; every variant that accepts the shape must agree with the oracle"""
    rng = np.random.default_rng(2024)
    sizes = [(3, 64), (64, 65), (65, 65), (5, 1024), (1025, 4), (127, 129), (33, 31), (256, 255), (513, 40),
             (4095, 5), (1000, 24), (700, 900), (2049, 2049), (263, 135)]       # incl. pairwise-sum depths 4 and 6
    for n1, n2 in sizes:
        s = n1 + n2 + int(rng.integers(0, 7))
        n = 97
        ps = synth.make_ps_matrix(n, s, seed=n1 * 7 + n2, nan_frac=float(rng.choice([0.0, 0.05, 0.5])))
        ps[3, :] = np.nan
        ps[4, :] = 1.0
        cols = rng.permutation(s)
        g1, g2 = np.sort(cols[:n1]), np.sort(cols[n1:n1 + n2])
        want = O.compare_rows(ps, g1, g2)
        big = max(n1, n2)
        variants = [0, 2] + ([1, 4] if big <= 64 else []) + ([3, 5] if big <= 1024 else [])
        for variant in variants:
            ctx.set_param("ranksum.variant", variant)
            try:
                got = ctx.ranksum(ps, g1, g2)
            finally:
                ctx.set_param("ranksum.variant", 0)
            _check_ranksum(got, want)


# ------------------------------------------------------------------------------ shape fuzz
def test_ps_fuzz_shapes(ctx):
    """random (n, s) incl. s not a multiple of 4, chunk boundaries around 256/128, heavy rows, far
    neighbours, count magnitudes across the 2^24 switch; PS and exclusion sums vs the oracle."""
    rng = np.random.default_rng(4242)
    shapes = [(1, 1), (2, 3), (37, 5), (500, 127), (500, 129), (300, 255), (300, 257), (200, 260), (150, 385),
              (64, 513), (900, 100), (400, 12), (90, 1030)]
    for n, s in shapes:
        deg = rng.integers(0, 9, size=n)
        if n > 20:
            deg[rng.integers(0, n, size=3)] = rng.integers(17, min(n, 60) + 1, size=3)     # heavy rows
        row_ptr = np.r_[0, np.cumsum(deg)].astype(np.int64)
        near = np.repeat(np.arange(n), deg) + rng.integers(-6, 7, size=int(row_ptr[-1]))
        far = rng.integers(0, n, size=near.size)
        col = np.clip(np.where(rng.random(near.size) < 0.9, near, far), 0, n - 1).astype(np.int32)
        scale = int(rng.choice([1, 1, 1000, 200000]))
        counts = (synth.make_counts(n, s, seed=n * 31 + s).astype(np.int64) * scale).clip(0, (1 << 24) - 1).astype(np.int32)
        want_ps, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
        ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
        assert np.array_equal(excl, want_excl), (n, s, scale)
        assert np.array_equal(ps, want_ps, equal_nan=True), (n, s, scale)


def test_fisher_fuzz_tables(ctx):
    """random 2x2 tables over many magnitudes incl. zero margins, a at either end of the support, and
    large balanced tables (mathematical ties across the mode); vs scipy through the oracle"""
    from scipy.stats import fisher_exact
    rng = np.random.default_rng(99)
    tabs = []
    for mag in (3, 30, 300, 3000, 100000):
        t = rng.integers(0, mag, size=(60, 4))
        t[rng.random(60) < 0.15, rng.integers(0, 4)] = 0
        tabs.append(t)
    sym = np.array([[k, m - k, m - k, k] for m in (10, 101, 1000, 5000) for k in (0, 1, m // 3, m // 2, m)])
    tabs.append(sym)
    tables = np.concatenate(tabs).astype(np.int64)
    want = np.array([fisher_exact([[a, b], [c, d]])[1] for a, b, c, d in tables])
    got = ctx.fisher_tables(tables)
    # margins of ~1e5: pmf(a) = exp(sum of nine log-factorials of size ~1e6) -- both scipy's and this
    # evaluation carry ~1e-9 relative error there; everything smaller agrees to 1e-9
    big = tables.sum(axis=1) > 50_000
    np.testing.assert_allclose(got[~big], want[~big], rtol=P_RTOL_TIGHT, atol=0)
    np.testing.assert_allclose(got[big], want[big], rtol=1e-7, atol=0)


@pytest.mark.parametrize("table,want,scipy_says", [
    ((41976, 5113, 372553, 78120), 6.575064524545e-310, 3.1204020472848e-310),
    ((58002, 90233, 44143, 91836), 3.450966061159082e-300, 1.7711130111622034e-300)])
def test_fisher_p_near_underflow_is_the_exact_sum(ctx, table, want, scipy_says):
    """p-values of 1e-300 and below: scipy's answer is erratic there -- Boost's hypergeometric pmf returns spurious
    zeros near the underflow limit (first table: pmf(36198) = 1.26e-310, pmf(36203) = 0, pmf(36205) = 3.0e-309; second
    table: pmf(48545) = 1.0e-300, pmf(48546) = 0, pmf(48547) = 1.8e-300), fisher_exact's boundary search is misled and
    the far tail drops out of the sum (scipy 1.15.3 returns `scipy_says`, the near tail alone) -- while the kernel's
    ratio walk never leaves the normal range until the final product.  The kernel returns the EXACT two-sided sum:
    `want` comes from rational arithmetic (tools/exact_fisher.py, ~5 min per table; both found by tests/fuzz_gpu.py)."""
    a, b, c, d = table
    got = ctx.fisher_tables(np.array([table], np.int64))[0]
    assert abs(got - want) <= 1e-7 * want                    # (pmf(a) = exp of nine log-factorials of ~6e6: ~1e-9)
    assert abs(got - scipy_says) > 0.4 * want
    pair = ctx.fisher_pairs(np.array([[a, b]], np.int32), np.array([[c, d]], np.int64))[0, 0]      # the pair kernel
    assert abs(pair - want) <= 1e-7 * want


def test_cluster_fuzz(ctx):
    """random junction sets: tiny, one chromosome, identical coordinates on both strands, heavy overlap"""
    rng = np.random.default_rng(31337)
    for n, n_chrom, span in [(1, 1, 10), (2, 1, 10), (3, 2, 1000), (400, 1, 300), (700, 5, 50), (1500, 2, 20000)]:
        cr = rng.integers(0, n_chrom, size=n).astype(np.int32)
        left = rng.integers(0, 5000, size=n).astype(np.int32)
        right = (left + rng.integers(1, span + 1, size=n)).astype(np.int32)
        strand = rng.integers(0, 2, size=n).astype(np.int8)
        key = np.unique(np.stack([cr, left, right, strand], axis=1), axis=0)      # the reference holds a SET
        cr, left, right, strand = (key[:, 0].astype(np.int32), key[:, 1].astype(np.int32), key[:, 2].astype(np.int32),
                                   key[:, 3].astype(np.int8))
        perm = rng.permutation(cr.size)
        cr, left, right, strand = cr[perm], left[perm], right[perm], strand[perm]
        want = O.cluster_csr(cr, left, right, strand)
        got = ctx.cluster(cr, left, right, strand)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), (n, n_chrom, span)


def test_bh_fuzz(ctx):
    """vector and per-column BH over awkward sizes (1, 2, tile boundaries of the radix sort), heavy ties,
    all ones, denormal-small p-values"""
    rng = np.random.default_rng(515)
    for m in (1, 2, 3, 255, 256, 257, 3071, 3072, 3073, 6145, 40_000):
        p = rng.random(m) ** rng.choice([1, 4, 30])
        p[rng.random(m) < 0.3] = 1.0
        if m > 10:
            p[:5] = [0.0, 5e-324, 1e-300, 1.0, 0.5]
            p[5:10] = p[10]                      # exact ties
        np.testing.assert_allclose(ctx.bh(p), O.bh_fdr(p), rtol=1e-14, atol=0)
    for n, cols in ((1, 1), (1, 7), (2, 3), (257, 2), (3073, 5), (50, 300), (700, 33)):
        p = rng.random((n, cols)) ** 3
        p[rng.random((n, cols)) < 0.2] = 1.0
        np.testing.assert_allclose(ctx.bh_columns(p), O.bh_columns(p), rtol=1e-14, atol=0)


@pytest.mark.parametrize("n", [16384, 16385, 100_000, 1_000_000, 2_097_152])
def test_bh_vector_samplesort_vs_radix(ctx, n):
    """the five-launch sample-sort path for one long vector (bh_cols.hip, bhv_*) against the radix path, bit for bit,
    and the oracle: continuous values, heavy ties (a tenth exactly 1, a block of one value, values ulps apart), zeros,
    denormals, NaN; then the masked variant (absent entries by flag and by negative p)"""
    rng = np.random.default_rng(n)
    p = rng.random(n) ** rng.choice([1, 3, 20])
    p[rng.random(n) < 0.1] = 1.0
    p[100:4100] = 0.25                                           # one value, a whole bucket's worth
    p[5000:5400] = 1.0 - rng.integers(1, 4, size=400) * 2.0 ** -53
    p[6000:6005] = [0.0, 5e-324, 1e-300, 1.0, 0.5]
    p = p[rng.permutation(n)]
    out = {}
    try:
        for path in (1, 2):
            ctx.set_param("bh.vector_path", path)
            d_p, d_q = ctx.to_device(p), ctx.empty(n, np.float64)
            ctx.bh_dev(d_p, d_q)
            out[path] = d_q.to_host()
        assert np.array_equal(out[1].view(np.uint64), out[2].view(np.uint64))
        if n <= 100_000:
            np.testing.assert_allclose(out[2], O.bh_fdr(p), rtol=1e-14, atol=0)
        # NaN sorts behind every number in both paths
        pn = p.copy()
        pn[rng.integers(0, n, size=3)] = np.nan
        for path in (1, 2):
            ctx.set_param("bh.vector_path", path)
            d_p, d_q = ctx.to_device(pn), ctx.empty(n, np.float64)
            ctx.bh_dev(d_p, d_q)
            out[path] = d_q.to_host()
        assert np.array_equal(out[1].view(np.uint64), out[2].view(np.uint64))
        # masked: a third of the entries absent
        tested = (rng.random(n) < 0.67).astype(np.uint8)
        for path in (1, 2):
            ctx.set_param("bh.vector_path", path)
            d_p, d_t, d_q = ctx.to_device(p), ctx.to_device(tested), ctx.empty(n, np.float64)
            ctx.bh_masked_dev(d_p, d_t, d_q)
            out[path] = d_q.to_host()
        assert np.array_equal(out[1].view(np.uint64), out[2].view(np.uint64))
        assert not out[2][tested == 0].any()
        if n <= 100_000:
            np.testing.assert_allclose(out[2][tested != 0], O.bh_fdr(p[tested != 0]), rtol=1e-14, atol=0)
        pm = np.where(tested != 0, p, -1.0)                      # absent = negative p, no flag array
        for path in (1, 2):
            ctx.set_param("bh.vector_path", path)
            d_p, d_q = ctx.to_device(pm), ctx.empty(n, np.float64)
            ctx.bh_masked_dev(d_p, None, d_q)
            out[path] = d_q.to_host()
        assert np.array_equal(out[1].view(np.uint64), out[2].view(np.uint64))
        if n <= 100_000:
            # buckets beyond the LDS capacity of a bucket workgroup (forced: every bucket) take the in-HBM network
            ctx.set_param("bh.vector_path", 2)
            ctx.set_param("bhv.cap", 512)
            d_p, d_q = ctx.to_device(pm), ctx.empty(n, np.float64)
            ctx.bh_masked_dev(d_p, None, d_q)
            assert np.array_equal(out[1].view(np.uint64), d_q.to_host().view(np.uint64))
    finally:
        ctx.set_param("bh.vector_path", 0)
        ctx.set_param("bhv.cap", 5632)


def test_bh_high_word_runs(ctx):
    """keys that agree in their high 32 bits: short runs of distinct values, long runs of one value, long
    runs of distinct values (p-values that differ by a few ulps, as Fisher's "almost 1" results do)"""
    rng = np.random.default_rng(808)
    m = 5000
    p = rng.random(m) ** 2
    p[100:110] = 0.25 + np.arange(10)[::-1] * 1e-13
    p[200:1200] = 1.0
    p[1300:1330] = 0.125 + rng.permutation(30) * 1e-14
    p[2000:2100] = 0.5 + rng.permutation(100) * 1e-12
    p[3000:3400] = 1.0 - rng.integers(1, 4, size=400) * 2.0 ** -53
    vec = p[rng.permutation(m)]
    np.testing.assert_allclose(ctx.bh(vec), O.bh_fdr(vec), rtol=1e-14, atol=0)
    cols = np.stack([p[rng.permutation(m)], p[rng.permutation(m)], rng.random(m)], axis=1)
    np.testing.assert_allclose(ctx.bh_columns(cols), O.bh_columns(cols), rtol=1e-14, atol=0)


# ------------------------------------------------------------------------------ row statistics (findOutliers)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rowstats_bit_identical_to_numpy(ctx, dtype):
    """np.nanmean / np.nanstd per row over a column subset, in the matrix dtype, bit for bit"""
    import warnings
    rng = np.random.default_rng(1234)
    for s, k in ((12, 10), (40, 12), (130, 100), (300, 129), (700, 500), (1100, 972), (1030, 1024), (9, 1)):
        n = 200
        data = np.round(rng.random((n, s)), 3).astype(dtype)
        data[rng.random((n, s)) < 0.15] = np.nan
        data[0, :] = np.nan
        data[1, :] = 0.25
        idx = np.sort(rng.permutation(s)[:k]).astype(np.int32)
        mean, std, n_nan = ctx.rowstats(data, idx)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_mean = np.array([np.nanmean(r[idx]) for r in data], dtype=dtype)
            want_std = np.array([np.nanstd(r[idx]) for r in data], dtype=dtype)
        assert mean.dtype == dtype and std.dtype == dtype
        assert np.array_equal(n_nan, np.isnan(data[:, idx]).sum(axis=1))
        assert np.array_equal(mean, want_mean, equal_nan=True), (s, k)
        assert np.array_equal(std, want_std, equal_nan=True), (s, k)


def test_ranksum_counting_and_sorting_rows_mixed(ctx):
    """groups of 65..1024: rows of 3-decimal values take the histogram kernel, any other row the sorting
    kernel, within one call; both must agree with the oracle (and with the sorting kernel alone)"""
    rng = np.random.default_rng(77)
    for n1, n2 in ((70, 90), (500, 500), (1024, 130)):
        s, n = n1 + n2, 150
        ps = synth.make_ps_matrix(n, s, seed=n1 + n2, nan_frac=0.07)          # 3-decimal values
        odd = rng.choice(n, size=40, replace=False)
        ps[odd[:20], 5] = np.float32(0.1234567)                              # one non-quantised value in the row
        ps[odd[20:]] = rng.random((20, s)).astype(np.float32)                # fully continuous rows
        ps[odd[25], :] = np.nan
        ps[3, : s - 2] = np.nan                                              # untested
        ps[4, :] = 0.0
        ps[5, :] = 1.0
        ps[6, :n1] = 0.0
        ps[6, n1:] = 1.0                                                     # complete separation
        g1, g2 = np.arange(n1, dtype=np.int32), np.arange(n1, s, dtype=np.int32)
        want = O.compare_rows(ps, g1, g2)
        for variant in (0, 3, 5):
            ctx.set_param("ranksum.variant", variant)
            try:
                got = ctx.ranksum(ps, g1, g2)
            finally:
                ctx.set_param("ranksum.variant", 0)
            _check_ranksum(got, want)


def test_chi2_fuzz_magnitudes(ctx):
    """Yates chi-square p-values over several count magnitudes (tiny tables where the correction clips
    the statistic to zero, large tables where p underflows towards 0)"""
    rng = np.random.default_rng(404)
    for mean_i, mean_e in ((1.5, 3), (8, 20), (200, 900), (20000, 50000)):
        incl = (rng.poisson(mean_i, size=(20, 7)) + 1).astype(np.int32)
        excl = (rng.poisson(mean_e, size=(20, 7)) + 1).astype(np.int64)
        p, n_bad = ctx.chi2_pairs(incl, excl)
        assert n_bad == 0
        want = O.chi2_pairs(incl, excl)
        ok = want > 1e-290
        np.testing.assert_allclose(p[ok], want[ok], rtol=1e-8, atol=0)
        assert (p[~ok] <= 1e-280).all()


def test_rowstats_rejects_too_many_columns(ctx):
    from splicedice_amd.engine import SdiceError
    data = np.zeros((4, 1100), np.float32)
    with pytest.raises(SdiceError):
        ctx.rowstats(data, np.arange(1025))
    with pytest.raises(SdiceError):
        ctx.rowstats(data, np.array([1100]))
