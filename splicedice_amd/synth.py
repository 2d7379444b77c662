"""Seeded synthetic inputs for tests, golden fixtures and bench.py.

Shapes follow SURVEY.md section 8(d): junctions are laid out in "genes" so that the
interval-overlap graph the reference builds (SPLICEDICE.py:230-255) has a realistic
average degree (5-10), counts are negative-binomial with a zero fraction.

Everything is `numpy.random.default_rng(seed)` (PCG64), so the same seed gives the
same arrays on the build container and on the GPU box.
"""
import numpy as np

STRANDS = ("+", "-")  # strand code 0 = '+', 1 = '-'  ('+' < '-' as Python strings)


def chrom_names(n_chrom):
    """chr1..chrN; NOTE Python string order is chr1 < chr10 < chr2 (SPLICEDICE.py:96,237)."""
    return [f"chr{i + 1}" for i in range(n_chrom)]


def chrom_ranks(names):
    """Dense rank of each chromosome name under Python `sorted()` (string order)."""
    order = {c: i for i, c in enumerate(sorted(set(names)))}
    return order


def make_junctions(n, seed, n_chrom=24, gene_spacing=20000, max_in_gene=12,
                   left_jitter=5000, min_len=50, len_span=20000):
    """n unique junctions as (chrom_rank int32, left int32, right int32, strand int8).

    Returned in a seeded random order (the reference holds them in a Python set, i.e.
    unordered, SPLICEDICE.py:156).
    """
    rng = np.random.default_rng(seed)
    out = None
    want = n
    total = np.empty((0, 4), dtype=np.int64)
    while total.shape[0] < n:
        m = int(want * 1.05) + 64
        per_gene = rng.integers(2, max_in_gene + 1, size=m // 2 + 1)
        gene_of = np.repeat(np.arange(per_gene.size), per_gene)[:m]
        n_genes = int(gene_of[-1]) + 1
        g_chrom = rng.integers(0, n_chrom, size=n_genes)
        g_strand = rng.integers(0, 2, size=n_genes)
        # genes are placed on a per-(chrom,strand) grid of `gene_spacing`
        slot = np.zeros(n_genes, dtype=np.int64)
        key = g_chrom * 2 + g_strand
        order = np.argsort(key, kind="stable")
        ks = key[order]
        start = np.r_[0, np.flatnonzero(np.diff(ks)) + 1]
        cnt = np.diff(np.r_[start, ks.size])
        slot[order] = np.arange(ks.size) - np.repeat(start, cnt)
        g_start = 10000 + slot * gene_spacing
        left = g_start[gene_of] + rng.integers(0, left_jitter, size=m)
        length = min_len + rng.integers(0, len_span, size=m)
        block = np.stack([g_chrom[gene_of], left, left + length, g_strand[gene_of]], axis=1)
        total = np.unique(np.concatenate([total, block]), axis=0)
        want = n - total.shape[0] + 64
    perm = rng.permutation(total.shape[0])[:n]
    out = total[perm]
    return (out[:, 0].astype(np.int32), out[:, 1].astype(np.int32),
            out[:, 2].astype(np.int32), out[:, 3].astype(np.int8))


def make_counts(n, s, seed, mean=30.0, disp=0.5, zero_frac=0.2, chunk=1 << 18):
    """int32 [n, s] row-major; NegBin(mean, disp) clipped to [0, 2**24), `zero_frac` zeros.

    Counts stay below 2**24 so the reference's float32 count storage
    (SPLICEDICE.py:259) is exact and "bit-exact integer counts" is well defined.
    """
    rng = np.random.default_rng(seed)
    r = 1.0 / disp
    p = r / (r + mean)
    out = np.empty((n, s), dtype=np.int32)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        x = rng.negative_binomial(r, p, size=(b - a, s))
        z = rng.random((b - a, s), dtype=np.float32) < zero_frac
        x[z] = 0
        np.clip(x, 0, (1 << 24) - 1, out=x)
        out[a:b] = x
    return out


def make_counts_rows(lo, hi, s, seed, block=1 << 16, **kw):
    """rows [lo, hi) of a count table that is defined block by block (one seeded block of `block`
    rows each), so that any rank can generate exactly its own row range of ONE shared dataset."""
    out = np.empty((hi - lo, s), dtype=np.int32)
    for b in range(lo // block, (max(hi, lo + 1) - 1) // block + 1):
        a0, a1 = b * block, (b + 1) * block
        r0, r1 = max(lo, a0), min(hi, a1)
        if r1 > r0:
            blk = make_counts(block, s, (seed, b), **kw)
            out[r0 - lo: r1 - lo] = blk[r0 - a0: r1 - a0]
    return out


def chrom_ranges(chrom_rank, world):
    """contiguous chromosome ranges with near-equal junction counts -> [(c_lo, c_hi, row_lo, row_hi)] per rank:
    rows are in (chrom, ...) order, so a chromosome range is a row range and no overlap edge crosses it"""
    cnt = np.bincount(np.asarray(chrom_rank, dtype=np.int64))
    cum = np.concatenate([[0], np.cumsum(cnt)])
    n = int(cum[-1])
    bounds = [0]
    for k in range(1, world):
        j = int(np.argmin(np.abs(cum - k * n / world)))
        bounds.append(max(j, bounds[-1]))
    bounds.append(cnt.size)
    return [(bounds[k], bounds[k + 1], int(cum[bounds[k]]), int(cum[bounds[k + 1]])) for k in range(world)]


def make_ps_matrix(n, s, seed, shift_frac=0.05, nan_frac=0.05, g1=None):
    """float32 [n, s] PS-like table for compare_sample_sets.

    per-row base ~ Beta(0.5, 0.5); the second half of the columns is shifted for
    `shift_frac` of the rows; noise N(0, 0.1); clipped to [0, 1]; rounded to three
    decimals exactly as the `_allPS.tsv` text round trip does (SPLICEDICE.py:353 ->
    compareSampleSets.py:202); `nan_frac` NaNs.
    """
    rng = np.random.default_rng(seed)
    half = s // 2 if g1 is None else g1
    out = np.empty((n, s), dtype=np.float32)
    chunk = 1 << 17
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        base = rng.beta(0.5, 0.5, size=(b - a, 1))
        shift = np.where(rng.random((b - a, 1)) < shift_frac,
                         rng.uniform(-0.4, 0.4, size=(b - a, 1)), 0.0)
        x = base + rng.normal(0.0, 0.1, size=(b - a, s))
        x[:, half:] += shift
        np.clip(x, 0.0, 1.0, out=x)
        x = np.rint(x * 1000.0) / 1000.0
        x = x.astype(np.float32)
        x[rng.random((b - a, s)) < nan_frac] = np.nan
        out[a:b] = x
    return out


def junction_name(chrom, left, right, strand):
    return f"{chrom}:{left}-{right}:{strand}"


def write_c1_dataset(dirname, seed=1, n_target=1000, n_samples=4, sj_sample=True):
    """Config 1 (plumbing): a manifest-style set of junction files on disk.

    Returns [(sample_name, path)], n_samples entries.  Samples 0..n-2 are
    splicedicebed files (bam_to_junc_bed.py:232-233 line format), the last one is a
    STAR SJ.out.tab when `sj_sample`.  Chromosomes chr1, chr10, chr2 exercise the
    Python string ordering; about 10 % of lines are omitted per sample and some
    counts fall below the default --minUnique (5).
    """
    import os
    rng = np.random.default_rng(seed)
    names = ["chr1", "chr10", "chr2"]
    cr, left, right, strand = make_junctions(n_target, seed, n_chrom=3)
    n = cr.size
    files = []
    for si in range(n_samples):
        cnt = make_counts(n, 1, seed * 100 + si, zero_frac=0.15)[:, 0]
        low = rng.random(n) < 0.08
        cnt[low] = rng.integers(1, 5, size=int(low.sum()))
        keep = rng.random(n) > 0.10
        is_sj = sj_sample and si == n_samples - 1
        path = os.path.join(dirname, f"s{si}.SJ.out.tab" if is_sj else f"s{si}.junc.bed")
        with open(path, "w") as fh:
            for j in np.flatnonzero(keep):
                c, l, r, st = names[cr[j]], int(left[j]), int(right[j]), STRANDS[strand[j]]
                if is_sj:
                    motif = 1 if rng.random() > 0.05 else 0
                    uniq = int(cnt[j])
                    multi = int(rng.integers(0, 3))
                    fh.write(f"{c}\t{l + 1}\t{r}\t{1 if st == '+' else 2}\t{motif}\t1\t{uniq}\t{multi}\t30\n")
                else:
                    e1 = 0.6 + 1.4 * rng.random()
                    e2 = 0.6 + 1.4 * rng.random()
                    ov = int(rng.integers(3, 40))
                    ann = "?" if rng.random() > 0.3 else "GENE1"
                    fh.write(f"{c}\t{l}\t{r}\te:{e1:.2f}:{e2:.2f};o:{ov};m:GT_AG;a:{ann}\t{int(cnt[j])}\t{st}\n")
        files.append((f"s{si}", path))
    return files
