#!/usr/bin/env python3
"""Extended differential campaign: the HIP path against the oracle on random ADVERSARIAL inputs, for a time budget.

The `-m gpu` suite holds a fixed set of fuzz cases; this tool draws fresh ones until the budget is spent and stops at
the first mismatch, printing the generator + seed that produced it (re-run with --only GEN --seed S).  It is a checker
run by hand on the GPU box; it lives under tests/ because it uses the oracle (test infrastructure), like the suite:

    python tests/fuzz_gpu.py --seconds 300 [--seed 1] [--only cluster]

Generators
  cluster   junction sets that stress the sample sort and the neighbour walks: thousands of junctions sharing one
            (chrom, left), one giant junction spanning a chromosome, nested ladders, one strand only, runs of touching
            junctions (right == next left), sorted / reverse-sorted input order, 2 049..40 000 junctions (several buckets)
  ps        random CSR (near + far neighbours, heavy rows) over random shapes incl. chunked widths
  ps_f64    float64 table, sums in list order
  ranksum   random group sizes 3..130 (lane, lane-pair, wave kernels), NaN density, quantised and raw values
  fisher    random count rows through fisher_pairs (pair table, long walks) against scipy
  bh        per-column sample-sort path against the generic path (bit for bit) and the oracle: tie structures,
            sizes around the bucket limits
  bh_vector one long vector (16 384..2 Mi values): the sample-sort path against the radix path, bit for bit, plain and masked,
            tie structures, some buckets forced beyond their slot
  cluster_big  70 k..2.6 M skewed junctions (a tenth in one locus, 5 000 sharing one left end, unequal chromosomes):
            the fast path against the generic radix-sort path, both on the GPU
  chi2, quantize   --chi2 p-values against scipy; the '.3f' round trip around every rounding boundary
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle_np as O  # noqa: E402
from splicedice_amd import synth  # noqa: E402
from splicedice_amd.engine import Context  # noqa: E402


def expected_nnz(cr, left, right, strand):
    """number of neighbour-list entries of a junction set (two per overlapping pair), without building the lists"""
    g = cr.astype(np.int64) * 2 + strand
    o = np.lexsort((left, g))
    key = (g[o] << 40) + left[o]
    ub = np.searchsorted(key, (g[o] << 40) + right[o], side="right")
    return 2 * int((ub - np.arange(len(o)) - 1).sum())


def gen_cluster(ctx, rng):
    n = int(rng.choice([1, 2, 5, 300, 2049, 2500, 5000, 9000, 20000, 40000]))
    kind = int(rng.integers(0, 8))
    n_chrom = int(rng.choice([1, 2, 5, 30]))
    cr = rng.integers(0, n_chrom, size=n)
    left = rng.integers(0, max(10, n * int(rng.choice([1, 5, 50]))), size=n)
    ln = rng.integers(1, int(rng.choice([5, 200, 20000])) + 1, size=n)
    strand = rng.integers(0, 2, size=n)
    if kind == 1:                                   # thousands share one (chrom, left)
        k = min(n, int(rng.choice([600, 3000, 18000])))
        cr[:k] = cr[0]; left[:k] = left[0]; ln[:k] = np.arange(1, k + 1)
    elif kind == 2:                                 # one giant junction per chromosome
        for c in range(n_chrom):
            i = int(rng.integers(0, n))
            cr[i] = c; left[i] = 0; ln[i] = int(left.max()) + 30000
    elif kind == 3:                                 # nested ladder: [i, 2n - i]
        k = min(n, 3000)
        left[:k] = np.arange(k); ln[:k] = 2 * k - 2 * np.arange(k) + 1; cr[:k] = 0; strand[:k] = 0
    elif kind == 4:                                 # one strand only
        strand[:] = int(rng.integers(0, 2))
    elif kind == 5:                                 # touching chain: right == next left
        left = np.arange(n) * 7; ln[:] = 7; cr[:] = 0
    elif kind == 6:                                 # few distinct lefts, many rights
        left = rng.integers(0, 12, size=n) * 1000
        ln = rng.integers(1, 4000, size=n)
    key = np.unique(np.stack([cr, left, left + ln, strand], axis=1), axis=0)
    order = int(rng.integers(0, 3))
    if order == 0:
        key = key[rng.permutation(len(key))]
    elif order == 1:
        key = key[::-1]
    cr, left, right, strand = (np.ascontiguousarray(key[:, 0], np.int32), np.ascontiguousarray(key[:, 1], np.int32),
                               np.ascontiguousarray(key[:, 2], np.int32), np.ascontiguousarray(key[:, 3], np.int8))
    if expected_nnz(cr, left, right, strand) > 20_000_000:      # (the oracle holds the lists as Python objects)
        return None
    want = O.cluster_csr(cr, left, right, strand)
    got = ctx.cluster(cr, left, right, strand)
    for name, g, w in zip(("row_of", "row_ptr", "col"), got, want):
        if not np.array_equal(g, w):
            return f"cluster {name} differs (n={len(key)} kind={kind} order={order} n_chrom={n_chrom})"
    return None


def gen_cluster_big(ctx, rng):
    """several hundred buckets, skewed: the fast path against the generic (radix sort) path, both on the GPU"""
    n = int(rng.choice([70_000, 300_000, 1_100_000, 2_600_000]))
    kind = int(rng.integers(0, 5))
    n_chrom = int(rng.choice([1, 3, 24, 400]))
    cr = rng.integers(0, n_chrom, size=n)
    span = n * int(rng.choice([1, 20, 400]))
    left = rng.integers(0, span, size=n)
    ln = rng.integers(1, int(rng.choice([30, 3000, 40000])) + 1, size=n)
    strand = rng.integers(0, 2, size=n)
    if kind == 1:                                   # a tenth of all junctions in one narrow locus
        k = n // 10
        cr[:k] = 0; left[:k] = rng.integers(1000, 1000 + max(50, k // 20), size=k); ln[:k] = rng.integers(1, 500, size=k)
    elif kind == 2:                                 # 5 000 junctions with one (chrom, left)
        cr[:5000] = cr[0]; left[:5000] = left[0]; ln[:5000] = np.arange(1, 5001)
    elif kind == 3:                                 # chromosome sizes differ by 1000x
        cr = np.minimum(cr, rng.integers(0, n_chrom, size=n)) if n_chrom > 1 else cr
        cr[: n // 2] = 0
    elif kind == 4:
        strand[:] = 1
    key = (cr.astype(np.int64) << 40) | (left.astype(np.int64) << 1) | strand
    _, first = np.unique(np.stack([key, ln], axis=1), axis=0, return_index=True)
    first = first[rng.permutation(first.size)]
    cr, left, right, strand = (np.ascontiguousarray(cr[first], np.int32), np.ascontiguousarray(left[first], np.int32),
                               np.ascontiguousarray((left + ln)[first], np.int32), np.ascontiguousarray(strand[first], np.int8))
    if expected_nnz(cr, left, right, strand) > 400_000_000:
        # a dense draw (2.6 M junctions of up to 40 kb inside 2.6 Mb would be 5e10 list entries, 200 GB): the library must
        # refuse it cleanly under a cap, before it allocates anything of that size on either side
        from splicedice_amd.engine import SdiceError
        try:
            ctx.set_param("cluster.max_nnz", 400_000_000)
            ctx.cluster(cr, left, right, strand)
            return f"cluster_big: a list of > 4e8 entries was materialised under cluster.max_nnz = 4e8 (n={first.size} kind={kind})"
        except SdiceError as e:
            return None if "neighbour list" in str(e) else f"cluster_big: unexpected error for a dense draw: {e}"
        finally:
            ctx.set_param("cluster.max_nnz", 0)
    fast = ctx.cluster(cr, left, right, strand)
    try:
        ctx.set_param("cluster.generic", 1)
        generic = ctx.cluster(cr, left, right, strand)
    finally:
        ctx.set_param("cluster.generic", 0)
    for name, g, w in zip(("row_of", "row_ptr", "col"), fast, generic):
        if not np.array_equal(g, w):
            return f"cluster_big {name}: fast path != generic path (n={first.size} kind={kind} n_chrom={n_chrom} span={span})"
    return None


def gen_chi2(ctx, rng):
    s = int(rng.choice([2, 5, 17]))
    n = int(rng.choice([1, 30, 200]))
    mag = int(rng.choice([5, 80, 5000, 2_000_000]))
    incl = rng.integers(1, mag + 1, size=(n, s)).astype(np.int32)        # (a zero expected frequency aborts the run)
    excl = rng.integers(1, mag * int(rng.choice([1, 9])) + 1, size=(n, s)).astype(np.int64)
    want = O.chi2_pairs(incl, excl)
    got, bad = ctx.chi2_pairs(incl, excl)
    want = want[0] if isinstance(want, tuple) else want
    ok = want > 1e-290                               # (erfc / exp tails: 1e-8 as in the suite; underflowing p by magnitude)
    if bad or not np.allclose(got[ok], want[ok], rtol=1e-8, atol=0) or (got[~ok] > 1e-280).any():
        return f"chi2 differs (n={n} s={s} mag={mag} bad={bad})"
    return None


def gen_quantize(ctx, rng):
    m = int(rng.choice([1, 1000, 300_000]))
    x = rng.random(m).astype(np.float32)
    k = rng.integers(0, 1001, size=m)
    near = (k / 1000.0 + rng.choice([-1, 0, 1], size=m) * 0.0005).astype(np.float32)
    x = np.where(rng.random(m) < 0.5, x, np.nextafter(near, np.float32(rng.choice([0.0, 1.0]))))
    x[rng.random(m) < 0.05] = np.nan
    want = O.quantize3(x)
    got = ctx.quantize3(x.copy())
    same = np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    return None if same else f"quantize3 differs (m={m})"


def gen_ps(ctx, rng):
    n = int(rng.choice([1, 7, 150, 900, 4000]))
    s = int(rng.choice([1, 3, 100, 127, 129, 256, 257, 500, 1000]))
    if n * s > 1_200_000:
        n = max(1, 1_200_000 // s)
    deg = rng.integers(0, 12, size=n)
    if n > 30:
        deg[rng.integers(0, n, size=3)] = rng.integers(17, min(n, 300) + 1, size=3)
    row_ptr = np.r_[0, np.cumsum(deg)].astype(np.int64)
    near = np.repeat(np.arange(n), deg) + rng.integers(-40, 41, size=int(row_ptr[-1]))
    far = rng.integers(0, n, size=near.size)
    col = np.clip(np.where(rng.random(near.size) < 0.85, near, far), 0, n - 1).astype(np.int32)
    scale = int(rng.choice([1, 1, 1000, 200000]))
    counts = (synth.make_counts(n, s, seed=int(rng.integers(1 << 30))).astype(np.int64) * scale).clip(0, (1 << 24) - 1).astype(np.int32)
    want_ps, want_excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    ps, excl = ctx.ps(counts, row_ptr, col, want_excl=True)
    if not np.array_equal(excl, want_excl):
        return f"ps excl differs (n={n} s={s} scale={scale})"
    if not np.array_equal(ps, want_ps, equal_nan=True):
        return f"ps differs (n={n} s={s} scale={scale})"
    return None


def gen_ps_f64(ctx, rng):
    n = int(rng.choice([1, 9, 200, 1500]))
    s = int(rng.choice([1, 5, 64, 65, 200, 700]))
    n_out = n - int(rng.integers(0, n // 3 + 1))
    if n_out < 1:
        n_out = n
    deg = rng.integers(0, 14, size=n_out)
    row_ptr = np.r_[0, np.cumsum(deg)].astype(np.int64)
    col = rng.integers(0, n, size=int(row_ptr[-1])).astype(np.int32)
    counts = rng.gamma(0.5, 40.0, size=(n, s)) * 10.0 ** rng.integers(-6, 7, size=(n, 1))
    counts[rng.random((n, s)) < 0.25] = 0.0
    want = O.write_ps_values_f64(counts, row_ptr, col, n_out)
    got = ctx.ps_f64(counts, row_ptr, col, n_out=n_out)
    same = np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    return None if same else f"ps_f64 differs (n={n} n_out={n_out} s={s})"


def gen_ranksum(ctx, rng):
    n1, n2 = int(rng.integers(3, 131)), int(rng.integers(3, 131))
    n = int(rng.choice([1, 31, 32, 33, 97, 400]))
    u = rng.random()
    if u < 0.2:                                      # counting / wave kernels (65..1024), block kernel beyond
        n1, n2 = int(rng.integers(3, 1100)), int(rng.integers(60, 1100))
        n = int(rng.choice([1, 9, 40]))
    elif u < 0.27:
        n1, n2 = int(rng.integers(1000, 4097)), int(rng.integers(3, 4097))
        n = int(rng.choice([1, 7]))
    s = n1 + n2 + int(rng.integers(0, 9))
    ps = synth.make_ps_matrix(n, s, seed=int(rng.integers(1 << 30)), nan_frac=float(rng.choice([0.0, 0.05, 0.5, 0.95])))
    if rng.random() < 0.3:                           # raw (non-quantised) values in some rows -> redo path
        rows = rng.integers(0, n, size=max(1, n // 5))
        ps[rows] = rng.random((len(rows), s)).astype(np.float32)
    if rng.random() < 0.3:
        ps[rng.integers(0, n)] = np.float32(rng.choice([0.0, 1.0, 0.5]))
    cols = rng.permutation(s)
    g1, g2 = np.sort(cols[:n1]), np.sort(cols[n1:n1 + n2])
    want = O.compare_rows(ps, g1, g2)
    got = ctx.ranksum(ps, g1, g2)
    tested_w = want["tested"].astype(bool)
    if not np.array_equal(got["tested"].astype(bool), tested_w):
        return f"ranksum tested mask differs (n={n} n1={n1} n2={n2})"
    for k in ("med1", "med2", "mean1", "mean2", "delta", "z"):
        if not np.array_equal(np.asarray(got[k])[tested_w], np.asarray(want[k])[tested_w]):
            return f"ranksum {k} differs (n={n} n1={n1} n2={n2})"
    if not np.allclose(np.asarray(got["p"])[tested_w], np.asarray(want["p"])[tested_w], rtol=1e-9, atol=0):
        return f"ranksum p differs (n={n} n1={n1} n2={n2})"
    return None


def gen_fisher(ctx, rng):
    s = int(rng.choice([2, 3, 9, 24]))
    n = int(rng.choice([1, 17, 120]))
    mag = int(rng.choice([3, 40, 600, 8000, 120000]))
    incl = rng.integers(0, mag, size=(n, s)).astype(np.int32)
    excl = rng.integers(0, mag * int(rng.choice([1, 7])), size=(n, s)).astype(np.int64)
    z = rng.random((n, s))
    incl[z < 0.15] = 0
    excl[(z > 0.1) & (z < 0.25)] = 0
    want = O.fisher_pairs(incl, excl)
    got = ctx.fisher_pairs(incl, excl)
    rtol = 1e-9 if mag <= 8000 else 1e-7
    # near the underflow limit scipy's own answer is erratic (Boost's hypergeometric pmf returns spurious zeros for
    # values of 1e-300 and below -- intermediate products underflow -- and misleads fisher_exact's boundary search:
    # tests/test_gpu_parity.py test_fisher_p_near_underflow_is_the_exact_sum); those cells are compared by magnitude only
    tiny = (want < 1e-280) | (got < 1e-280)
    if (np.maximum(want, got)[tiny] > 1e-270).any():
        return f"fisher: one side only is near the underflow limit (n={n} s={s} mag={mag})"
    got, want = np.where(tiny, 1.0, got), np.where(tiny, 1.0, want)
    if not np.allclose(got, want, rtol=rtol, atol=0):
        bad = np.argwhere(~np.isclose(got, want, rtol=rtol, atol=0))[0]
        return f"fisher differs at {tuple(bad)}: {got[tuple(bad)]!r} vs {want[tuple(bad)]!r} (n={n} s={s} mag={mag})"
    return None


def gen_bh(ctx, rng):
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1023, 1025, 4097, 20000, 70000]))
    cols = int(rng.choice([1, 2, 7, 8, 9, 40]))
    if n * cols > 1_500_000:
        cols = max(1, 1_500_000 // n)
    p = rng.random((n, cols)) ** float(rng.choice([1, 3, 20]))
    kind = int(rng.integers(0, 11))
    if kind == 1:
        p[rng.random((n, cols)) < 0.6] = 1.0
    elif kind == 10:                                  # the top of a Fisher column: exact ones, a crowd of distinct sums a few
        q = rng.random((n, cols))                     # hundred ulps below 1 (heavy-tailed), ordinary values
        p[q < 0.06] = 1.0
        m = (q >= 0.06) & (q < 0.2)
        p[m] = 1.0 - (np.abs(rng.standard_cauchy(int(m.sum()))) * 30 + 1).astype(np.int64).clip(1, 10 ** 6) * 2.0 ** -53
    elif kind == 6:                                   # a dense cluster of DISTINCT values inside a wide range: one bin of the
        m = rng.random((n, cols)) < 0.5               # bucket's counting sort holds many different keys
        p[m] = 0.3 + rng.integers(0, 4000, size=int(m.sum())) * 2.0 ** -52
    elif kind == 7:                                   # p-values over hundreds of binades (key space is logarithmic there)
        p[:] = 10.0 ** -(rng.random((n, cols)) * 300)
    elif kind == 8:                                   # NaN, inf, negative values and zeros among ordinary ones
        q = rng.random((n, cols))
        p[q < 0.02] = np.nan
        p[(q >= 0.02) & (q < 0.03)] = np.inf
        p[(q >= 0.03) & (q < 0.05)] = -p[(q >= 0.03) & (q < 0.05)]
        p[(q >= 0.05) & (q < 0.08)] = 0.0
    elif kind == 9:                                   # two values only, one of them rare
        p[:] = np.where(rng.random((n, cols)) < 0.01, 0.25, 1.0)
    elif kind == 2:
        p[:] = rng.choice([1.0, 0.5, 0.0286, 0.2, 1e-5], size=(n, cols))
    elif kind == 3:
        p[:, 0] = 0.125
    elif kind == 4:
        p[:] = 0.5 + rng.integers(0, 9, size=(n, cols)) * 2.0 ** -53
    elif kind == 5:
        p[:] = np.sort(p, axis=0)[::-1] if rng.random() < 0.5 else np.sort(p, axis=0)
    wg, mean, fused = int(rng.choice([256, 256, 512, 1024])), int(rng.choice([0, 0, 100, 700, 2500])), int(rng.integers(0, 2))
    big_wg = int(rng.choice([256, 512]))
    try:
        ctx.set_param("bh.columns_path", 1)
        d = ctx.to_device(p); ctx.bh_columns_dev(d); generic = d.to_host()
        ctx.set_param("bh.columns_path", 2)
        ctx.set_param("bh.wg", wg); ctx.set_param("bh.mean", mean); ctx.set_param("bh.fused_count", fused); ctx.set_param("bh.big_wg", big_wg)
        d = ctx.to_device(p); ctx.bh_columns_dev(d); fast = d.to_host()
    finally:
        ctx.set_param("bh.columns_path", 0)
        ctx.set_param("bh.wg", 256); ctx.set_param("bh.mean", 0); ctx.set_param("bh.fused_count", 1); ctx.set_param("bh.big_wg", 512)
    if not np.array_equal(generic, fast, equal_nan=True):
        return f"bh columns: sample-sort path != generic path (n={n} cols={cols} kind={kind} wg={wg} mean={mean} fused={fused} big_wg={big_wg})"
    if kind == 8:
        return None                                   # (the oracle's definition is for proper p-values)
    if n <= 5000 and not np.allclose(fast, O.bh_columns(p), rtol=1e-14, atol=0):
        return f"bh columns differs from the oracle (n={n} cols={cols} kind={kind})"
    v = p[:, 0].copy()
    if not np.allclose(ctx.bh(v), O.bh_fdr(v), rtol=1e-14, atol=0):
        return f"bh vector differs (m={n} kind={kind})"
    return None


def gen_bh_vector(ctx, rng):
    """one long vector: the sample-sort path (bhv_*) against the radix path, bit for bit, plain and masked"""
    n = int(rng.choice([16384, 16385, 20000, 65536, 100_003, 262_144, 700_001, 1_048_576, 2_097_152]))
    p = rng.random(n) ** float(rng.choice([1, 3, 20]))
    kind = int(rng.integers(0, 7))
    if kind == 1:
        p[rng.random(n) < 0.6] = 1.0
    elif kind == 2:
        p[:] = rng.choice([1.0, 0.5, 0.0286, 0.2, 1e-5], size=n)
    elif kind == 3:
        p[:] = 0.125                                          # ONE value: every split is by index
    elif kind == 4:
        p[:] = 0.5 + rng.integers(0, 9, size=n) * 2.0 ** -53
    elif kind == 5:
        p[:] = np.sort(p)[::-1] if rng.random() < 0.5 else np.sort(p)
    elif kind == 6:
        p[: n // 3] = 0.0
        p[n // 3: n // 2] = 5e-324
    frac = float(rng.choice([1.0, 0.95, 0.5, 0.02]))
    tested = (rng.random(n) < frac).astype(np.uint8)
    cap = int(rng.choice([5632, 5632, 5632, 3072]))           # a small capacity sends some buckets through the slow path
    out = {}
    try:
        for path in (1, 2):
            ctx.set_param("bh.vector_path", path)
            ctx.set_param("bhv.cap", cap)
            d_p, d_q = ctx.to_device(p), ctx.empty(n, np.float64)
            ctx.bh_dev(d_p, d_q)
            a = d_q.to_host()
            d_t = ctx.to_device(tested)
            ctx.bh_masked_dev(d_p, d_t, d_q)
            out[path] = (a, d_q.to_host())
    finally:
        ctx.set_param("bh.vector_path", 0)
        ctx.set_param("bhv.cap", 5632)
    for k, what in ((0, "plain"), (1, "masked")):
        if not np.array_equal(out[1][k].view(np.uint64), out[2][k].view(np.uint64)):
            return f"bh vector ({what}): sample-sort path != radix path (n={n} kind={kind} tested={frac} cap={cap})"
    if n <= 70000 and not np.allclose(out[2][0], O.bh_fdr(p), rtol=1e-14, atol=0):
        return f"bh vector differs from the oracle (n={n} kind={kind})"
    return None


GENERATORS = {"cluster": gen_cluster, "ps": gen_ps, "ps_f64": gen_ps_f64, "ranksum": gen_ranksum, "fisher": gen_fisher,
              "bh": gen_bh, "bh_vector": gen_bh_vector, "cluster_big": gen_cluster_big, "chi2": gen_chi2, "quantize": gen_quantize}


def rss_gib():
    with open("/proc/self/status") as fh:
        for line in fh:
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / (1 << 20)
    return 0.0


CURRENT = {"what": "start"}


def watchdog(limit_gib, out):
    """a case that runs away with host memory ends the process here (exit code 3), long before the box's cap does"""
    import threading

    def loop():
        while True:
            time.sleep(0.1)
            g = rss_gib()
            if g > limit_gib:
                msg = f"RSS {g:.1f} GiB > {limit_gib} GiB during {CURRENT['what']}: aborting"
                print(msg, flush=True)
                if out:
                    with open(out, "a") as fh:
                        fh.write(msg + "\n")
                os._exit(3)
    threading.Thread(target=loop, daemon=True).start()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=None, help="progress / summary file (written as the run goes)")
    ap.add_argument("--verbose", action="store_true", help="one line per case (with the resident set size)")
    ap.add_argument("--first-case", type=int, default=0)
    ap.add_argument("--rss-limit-gib", type=float, default=40.0)
    a = ap.parse_args()
    watchdog(a.rss_limit_gib, a.out)
    names = [a.only] if a.only else list(GENERATORS)
    ctx = Context(0)
    t0 = time.time()
    done = {k: 0 for k in names}
    case = a.first_case
    last_note = t0
    failure = None
    while time.time() - t0 < a.seconds:
        name = names[case % len(names)]
        seed = a.seed * 1_000_003 + case
        rng = np.random.default_rng(seed)
        CURRENT["what"] = f"generator={name} seed={a.seed} case={case}"
        if a.verbose:
            line = f"case {case} {name} rss {rss_gib():.2f} GiB"
            print(line, flush=True)
            if a.out:
                with open(a.out, "a") as fh:
                    fh.write(line + "\n")
        with np.errstate(all="ignore"):
            msg = GENERATORS[name](ctx, rng)
        if msg:
            failure = f"MISMATCH generator={name} seed={a.seed} case={case}: {msg}"
            break
        done[name] += 1
        case += 1
        if time.time() - last_note > 30:
            last_note = time.time()
            line = f"[{time.time() - t0:6.0f}s] " + " ".join(f"{k}={v}" for k, v in done.items())
            print(line, flush=True)
            if a.out:
                with open(a.out, "a") as fh:
                    fh.write(line + "\n")
    ctx.close()
    summary = failure or ("OK " + " ".join(f"{k}={v}" for k, v in done.items()) + f" cases in {time.time() - t0:.0f}s, seed {a.seed}")
    print(summary, flush=True)
    if a.out:
        with open(a.out, "a") as fh:
            fh.write(summary + "\n")
    sys.exit(1 if failure else 0)


if __name__ == "__main__":
    main()
