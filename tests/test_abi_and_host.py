"""CPU: the C-ABI library loads and exports every symbol include/sdice.h declares (no compute
calls), fails loudly without a GPU, and the host-side logic (text parsing, CSR builders,
shard plan) behaves like the reference's host code."""
import argparse
import os
import re

import numpy as np

from oracle import oracle_quant_io as QIO
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(REPO, "include", "sdice.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdice_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from splicedice_amd import _ffi
    lib = _ffi.load()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdice.h but not exported"
        assert s in _ffi.SIGNATURES, f"{s} has no ctypes prototype in _ffi.SIGNATURES"
    assert set(_ffi.SIGNATURES) == set(syms)
    assert lib.sdice_version() == 1


def test_no_gpu_fails_loudly():
    """There is no CPU fallback: on a box without a HIP device context creation raises."""
    import ctypes
    from splicedice_amd import _ffi
    lib = _ffi.load()
    h = ctypes.c_void_p()
    rc = lib.sdice_ctx_create(0, ctypes.byref(h))
    if rc == 0:            # running on a GPU box: nothing to check here
        lib.sdice_ctx_destroy(h)
        pytest.skip("a HIP device is present")
    assert rc < 0 and h.value is None
    assert b"no CPU backend" in lib.sdice_last_error() or b"gfx950" in lib.sdice_last_error()
    from splicedice_amd.engine import Context, SdiceError
    with pytest.raises(SdiceError):
        Context(0)


def test_missing_library_message(monkeypatch, tmp_path):
    from splicedice_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_ffi.SdiceError, match="no CPU fallback"):
        _ffi.load()


# ----------------------------------------------------------------------------------- host logic
def _quant_args(**over):
    a = argparse.Namespace(maxLength=50000, minLength=50, minOverhang=5, drim=False, noMultimap=False,
                           filter="gtag_only", minUnique=5, lowCoverageNan=False, minEntropy=1)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def _manifest(golden_dir, tmp_path):
    qdir = os.path.join(golden_dir, "quant_c1")
    out = tmp_path / "manifest.tsv"
    with open(os.path.join(qdir, "manifest.rel.tsv")) as src, open(out, "w") as dst:
        for line in src:
            row = line.rstrip("\n").split("\t")
            row[1] = os.path.join(qdir, "inputs", row[1])
            dst.write("\t".join(row) + "\n")
    return str(out), qdir


@pytest.mark.parametrize("variant", ["default", "lowcov_drim", "strict"])
def test_quant_host_parsing_matches_reference_files(golden_dir, tmp_path, variant):
    """junction filters + count gathering (host Python) against the reference's _junctions.bed
    and _inclusionCounts.tsv; the sort here is Python's, the GPU sort is tested under -m gpu."""
    import json
    from splicedice_amd import quant, textio
    manifest_path, qdir = _manifest(golden_dir, tmp_path)
    exp = os.path.join(qdir, f"expected_{variant}")
    args = _quant_args(**json.load(open(os.path.join(exp, "args.json"))))
    manifest = quant.parse_manifest(manifest_path)
    assert [s.type for s in manifest] == ["splicedicebed", "splicedicebed", "splicedicebed", "SJ", "bed"]
    junctions = sorted(QIO.get_all_junctions(manifest, args))
    bed = [ln.split("\t")[3] for ln in open(os.path.join(exp, "out_junctions.bed"))]
    assert [textio.junction_name(j) for j in junctions] == bed
    index = {j: i for i, j in enumerate(junctions)}
    counts, low = QIO.get_junction_counts(manifest, index, args)
    with open(os.path.join(exp, "out_inclusionCounts.tsv")) as fh:
        fh.readline()
        for line, row in zip(fh, counts):
            assert line.rstrip("\n").split("\t")[1:] == [str(int(x)) for x in row]
    if args.lowCoverageNan:
        ps = [ln.rstrip("\n").split("\t")[1:] for ln in open(os.path.join(exp, "out_allPS.tsv"))][1:]
        flat = {int(i) for i in low}
        s = len(manifest)
        for i in flat:
            assert ps[i // s][i % s] == "nan"


def test_textio_roundtrip_and_ranks():
    from splicedice_amd import textio
    j = ("chr10", 5, 99, "-")
    assert textio.parse_junction_name(textio.junction_name(j)) == j
    names, cr, left, right, strand = textio.junction_arrays([("chr2", 1, 2, "+"), ("chr10", 3, 4, "-"), ("chr1", 5, 6, "+")])
    assert names == ["chr1", "chr10", "chr2"] and cr.tolist() == [2, 1, 0] and strand.tolist() == [0, 1, 0]
    with pytest.raises(ValueError):
        textio.junction_arrays([("chr1", 5, 2, "+")])
    with pytest.raises(ValueError):
        textio.counts_to_int32([[1.5]], "x")
    assert textio.counts_to_int32([["3", "0"]], "x").tolist() == [[3, 0]]


def test_pairwise_exclusion_csr_matches_isin_semantics():
    from splicedice_amd import pairwise
    events = ["a", "b", "c", "b"]                       # a repeated event name matches both rows
    clusters = {"a": ["b", "zzz", "b"], "b": ["a"], "c": []}
    row_ptr, col = pairwise.exclusion_csr(events, clusters)
    counts = np.arange(12).reshape(4, 3)
    for n, e in enumerate(events):
        want = counts[np.isin(events, clusters[e])].sum(axis=0)      # pairwise_fisher.py:158-160
        got = counts[col[row_ptr[n]:row_ptr[n + 1]]].sum(axis=0) if row_ptr[n + 1] > row_ptr[n] else np.zeros(3)
        assert np.array_equal(got, want)
    with pytest.raises(KeyError):
        pairwise.exclusion_csr(["nope"], clusters)


def test_pairwise_cluster_file_parsing(tmp_path):
    from splicedice_amd import pairwise
    f = tmp_path / "c.tsv"
    f.write_text("e1\te2,e3\ne4\t\ne5\n")
    assert pairwise.get_clusters(str(f)) == {"e1": ["e2", "e3"], "e4": [], "e5": []}


def test_compare_host_helpers(golden_dir):
    from splicedice_amd import compare_sample_sets as css
    d = os.path.join(golden_dir, "compare")
    rows, cols, m = css.read_ps_table(os.path.join(d, "in_allPS.tsv"))
    assert m.dtype == np.float32 and m.shape == (300, 12) and np.isnan(m[6, :6]).all()
    g2 = css.samples_from_manifest(os.path.join(d, "m2.tsv"))
    assert g2[0] == "samp11"
    assert css.column_indices(g2, cols).tolist() == [6, 7, 8, 9, 10, 11]          # table order, not manifest order
    assert css.column_indices(["samp0", "not_in_table"], cols).tolist() == [0]
    annotated, gene_coords, tids = css.read_annotation(os.path.join(d, "anno.gtf"))
    assert annotated[("chr1", 999, 2002, "+")] == ["GENEA"] and tids[("chr1", 999, 2002, "+")] == ["T1"]
    assert gene_coords[("chr1", "+")][(900, 2499)] == ["GENEA"]


def test_dispatcher_names_and_flags():
    from splicedice_amd.__main__ import build_parser
    p = build_parser()
    a = p.parse_args(["quant", "-m", "m", "-o", "o"])
    assert (a.maxLength, a.minLength, a.minOverhang, a.minUnique, a.minEntropy, a.filter) == (50000, 50, 5, 5, 1, "gtag_only")
    a = p.parse_args(["pairwise", "--inclusionSPLICEDICE", "c", "-c", "k"])
    assert a.multiple_test_correction == "pairwise" and a.output == "pairwise.tsv" and a.chi2 is False
    a = p.parse_args(["compare_sample_sets", "--psiSPLICEDICE", "p", "-m1", "a", "-m2", "b", "-o", "x"])
    assert a.annotation == ""
    a = p.parse_args(["counts_to_ps", "-i", "c", "-o", "o", "-r"])
    assert a.recluster and a.clusters is None
    a = p.parse_args(["findOutliers", "--psiSPLICEDICE", "m.npz", "-m", "man.tsv"])
    assert a.nullMan is None and a.outlierCutoff == 3 and a.dpsiThrsh == 0.1
    for name in ("bam_to_junc_bed", "intron_coverage", "subset", "select"):
        assert p.parse_args([name]).command == name
    a = p.parse_args(["ir_table", "-i", "c.tsv", "-c", "k.tsv", "-d", "cov", "-o", "out", "-r"])
    assert a.makeRSDtable and not a.allJunctions and not a.singleJunctionCalculation and a.RSDthreshold == 1.0
    a = p.parse_args(["similarity", "-c", "vs.tsv", "-a", "allps.tsv", "-o", "out.tsv"])
    assert a.manifest is None and a.comparison == "vs.tsv"


def test_console_entry_point_resolves():
    """pyproject.toml declares `splicedice = splicedice_amd.__main__:main`, the reference's console
    name (reference setup.py:200-204): the target must import and be the dispatcher's main."""
    import importlib
    import os
    try:
        import tomllib as toml          # Python >= 3.11
    except ImportError:
        import tomli as toml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as f:
        meta = toml.load(f)
    target = meta["project"]["scripts"]["splicedice"]
    modname, func = target.split(":")
    fn = getattr(importlib.import_module(modname), func)
    from splicedice_amd.__main__ import main
    assert fn is main
    assert "splicedice_amd" in meta["tool"]["setuptools"]["packages"]


# ----------------------------------------------------------------------------------- shard plan
def test_shard_plan_clean_cuts_and_halo():
    from oracle import oracle_np as O
    from splicedice_amd import shard, synth
    cr, l, r, st = synth.make_junctions(6000, 3, n_chrom=3)
    _, row_ptr, col = O.cluster_csr(cr, l, r, st)
    for world in (1, 2, 3, 8):
        plan = shard.shard_plan(row_ptr, col, world)
        assert plan[0]["own_lo"] == 0 and plan[-1]["own_hi"] == 6000
        assert all(plan[k]["own_hi"] == plan[k + 1]["own_lo"] for k in range(world - 1))
        sizes = [p["own_hi"] - p["own_lo"] for p in plan]
        assert max(sizes) - min(sizes) <= 0.1 * 6000 / world + 2
        for p in plan:      # gene-shaped data: clean cut nearby (zero halo) or a halo of a few rows
            assert 0 <= p["own_lo"] - p["ext_lo"] <= 64 and 0 <= p["ext_hi"] - p["own_hi"] <= 64
            rp, cl = shard.local_csr(row_ptr, col, p)
            a, b = p["own_lo"] - p["ext_lo"], p["own_hi"] - p["ext_lo"]
            seg = col[row_ptr[p["own_lo"]]:row_ptr[p["own_hi"]]]
            assert np.array_equal(cl[rp[a]:rp[b]], seg - p["ext_lo"])     # owned rows keep every neighbour
            assert cl.size == 0 or (cl.min() >= 0 and cl.max() < p["ext_hi"] - p["ext_lo"])
    assert any((p["ext_lo"], p["ext_hi"]) == (p["own_lo"], p["own_hi"]) for p in shard.shard_plan(row_ptr, col, 2))
    # one chain-linked block has no clean cut: the plan must fall back to halos
    n = 400
    row_ptr = np.arange(0, 2 * n + 1, 2, dtype=np.int64)
    col = np.stack([np.maximum(np.arange(n) - 1, 0), np.minimum(np.arange(n) + 1, n - 1)], axis=1).ravel().astype(np.int32)
    plan = shard.shard_plan(row_ptr, col, 4)
    assert [p["own_lo"] for p in plan] == [0, 100, 200, 300]
    assert plan[1]["ext_lo"] == 99 and plan[1]["ext_hi"] == 201
    rp, cl = shard.local_csr(row_ptr, col, plan[1])
    assert rp.size == 103 and cl.min() >= 0 and cl.max() <= 101


@pytest.mark.parametrize("variant", ["default", "nomulti_lowcov", "short"])
def test_quant_edge_cases_host_side(golden_dir, tmp_path, variant):
    """the hand-made edge-case inputs (tests/golden/quant_edge): the Python restatement of the parsing
    rules AND the library's C++ parser + union + lookup give the reference's junction list and counts"""
    import json
    from splicedice_amd import juncio, quant, textio
    qdir = os.path.join(golden_dir, "quant_edge")
    manifest_path = tmp_path / "manifest.tsv"
    with open(os.path.join(qdir, "manifest.rel.tsv")) as src, open(manifest_path, "w") as dst:
        for line in src:
            row = line.rstrip("\n").split("\t")
            row[1] = os.path.join(qdir, "inputs", row[1])
            dst.write("\t".join(row) + "\n")
    exp = os.path.join(qdir, f"expected_{variant}")
    args = _quant_args(**json.load(open(os.path.join(exp, "args.json"))))
    manifest = quant.parse_manifest(str(manifest_path))
    assert [s.type for s in manifest] == ["splicedicebed", "SJ", "bed", "bam", "leafcutter", "unknown"]
    bed = [ln.split("\t")[3] for ln in open(os.path.join(exp, "out_junctions.bed"))]
    with open(os.path.join(exp, "out_inclusionCounts.tsv")) as fh:
        fh.readline()
        want_counts = [line.rstrip("\n").split("\t")[1:] for line in fh]
    # (1) Python restatement
    junctions = sorted(QIO.get_all_junctions(manifest, args))
    assert [textio.junction_name(j) for j in junctions] == bed
    counts, _ = QIO.get_junction_counts(manifest, {j: i for i, j in enumerate(junctions)}, args)
    assert [[str(int(x)) for x in row] for row in counts] == want_counts
    # (2) C++ parser, host union, C++ row lookup
    names, junc, parsed = juncio.ingest(manifest, args, None)
    got_names = [f"{names[c]}:{l}-{r}:{'+-'[s]}" for c, l, r, s in zip(*junc)]
    assert got_names == bed
    counts2, _ = juncio.gather_counts(manifest, parsed, junc, args)
    assert [[str(int(x)) for x in row] for row in counts2] == want_counts


def test_union_key_decoding_above_2047_chromosomes():
    """ranks 2048..4095 set bit 63 of the packed key; they must not decode negative"""
    from splicedice_amd import juncio
    c = np.array([5, 2047, 2048, 4095], dtype=np.int64)
    l = np.array([10, 2 ** 31 - 1 - 1000, 7, 123456], dtype=np.int64)
    span = np.array([50, 999, (1 << 20) - 1, 0], dtype=np.int64)
    st = np.array([0, 1, 1, 0], dtype=np.int64)
    keys = ((c << 52) | (l << 21) | (span << 1) | st).astype(np.uint64)
    cr, left, right, strand = juncio._unpack_keys(keys)
    assert cr.tolist() == c.tolist() and left.tolist() == l.tolist()
    assert right.tolist() == (l + span).tolist() and strand.tolist() == st.tolist()


def test_strong_scaling_partition_helpers():
    from splicedice_amd import synth
    cr, l, r, st = synth.make_junctions(20000, 2)
    for world in (1, 2, 3, 8):
        rng_ = synth.chrom_ranges(cr, world)
        assert rng_[0][2] == 0 and rng_[-1][3] == cr.size
        for (c0, c1, r0, r1), nxt in zip(rng_, rng_[1:] + [None]):
            assert r1 - r0 == int(((cr >= c0) & (cr < c1)).sum())
            if nxt:
                assert nxt[0] == c1 and nxt[2] == r1
    a = synth.make_counts_rows(70000, 140001, 3, 5)
    b = synth.make_counts_rows(0, 140001, 3, 5)
    assert np.array_equal(a, b[70000:])


# ----------------------------------------------------------------------------------- ir_table (host logic)
class _SumEngine:
    """host double of the one engine call ir_table makes (exclusion sums over a CSR)"""

    def ps(self, counts, row_ptr, col, want_excl=False, want_ps=True):
        from oracle import oracle_np as O
        return O.calculate_psi_vectorised(counts, row_ptr, col)[1]


def _ir_args(golden_dir, tmp_path, tag, **kw):
    import argparse
    d = os.path.join(golden_dir, "ir_table")
    return argparse.Namespace(inclusionCounts=os.path.join(d, "in_inclusionCounts.tsv"), clusters=os.path.join(d, "in_allClusters.tsv"),
                              coverageDirectory=os.path.join(d, "coverage"), outputPrefix=str(tmp_path / tag), makeRSDtable=True,
                              annotation=os.path.join(d, "anno.gtf"), RSDthreshold=1.0, **kw)


def _table_by_column(path):
    with open(path) as fh:
        rows = [line.rstrip("\n").split("\t") for line in fh]
    return {(r[0], c): v for r in rows[1:] for c, v in zip(rows[0][1:], r[1:])}, [r[0] for r in rows[1:]], sorted(rows[0][1:])


@pytest.mark.parametrize("tag,kw", [("annotated", dict(allJunctions=False, singleJunctionCalculation=False)),
                                    ("all", dict(allJunctions=True, singleJunctionCalculation=False)),
                                    ("all_single", dict(allJunctions=True, singleJunctionCalculation=True))])
def test_ir_table_against_reference_goldens(golden_dir, tmp_path, tag, kw, capsys):
    """tests/golden/ir_table was written by the reference's ir_table.py (make_golden_ir.py).  Sample columns
    come in os.listdir order there and here, so cells are compared by (junction, sample); the message lines
    ("mxCluster ...") must be the same multiset."""
    from splicedice_amd import ir_table
    ir_table.run_with(_ir_args(golden_dir, tmp_path, tag, **kw), ctx=_SumEngine())
    said = [ln for ln in capsys.readouterr().out.splitlines() if not ln.startswith("Done")]
    d = os.path.join(golden_dir, "ir_table")
    with open(os.path.join(d, f"expected_{tag}_stdout.txt")) as fh:
        assert sorted(said) == sorted(fh.read().splitlines())
    for suffix in ("_intron_retention.tsv", "_intron_retention_RSD.tsv"):
        got, got_rows, got_cols = _table_by_column(str(tmp_path / tag) + suffix)
        want, want_rows, want_cols = _table_by_column(os.path.join(d, f"expected_{tag}{suffix}"))
        assert got_rows == want_rows and got_cols == want_cols and got == want


def test_ir_table_without_rsd_table_dies_like_the_reference(golden_dir, tmp_path):
    from splicedice_amd import ir_table
    a = _ir_args(golden_dir, tmp_path, "x", allJunctions=True, singleJunctionCalculation=False)
    a.makeRSDtable = False
    with pytest.raises(KeyError):          # RSD[sample][junction] is never filled without -r (ir_table.py:140-145)
        ir_table.run_with(a, ctx=_SumEngine())


def _shard_plan_numpy(row_ptr, col, world, max_shift_frac=0.02):
    """the round-1 numpy formulation of the plan (checker for sdice_shard_plan)"""
    from splicedice_amd import shard
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    n = row_ptr.size - 1
    clean_pos = np.flatnonzero(shard.clean_cuts(row_ptr, col))
    bounds = [0]
    for k in range(1, world):
        ideal = (k * n) // world
        j = np.searchsorted(clean_pos, ideal)
        cands = clean_pos[max(0, j - 1): j + 1]
        best = int(cands[np.argmin(np.abs(cands - ideal))]) if cands.size else ideal
        if abs(best - ideal) > max(1, int(max_shift_frac * n / world)):
            best = ideal
        bounds.append(max(best, bounds[-1]))
    bounds.append(n)
    plan = []
    for k in range(world):
        lo, hi = bounds[k], bounds[k + 1]
        ext_lo, ext_hi = lo, hi
        if hi > lo:
            seg = col[row_ptr[lo]:row_ptr[hi]]
            if seg.size:
                ext_lo, ext_hi = min(lo, int(seg.min())), max(hi, int(seg.max()) + 1)
        plan.append(dict(own_lo=lo, own_hi=hi, ext_lo=ext_lo, ext_hi=ext_hi))
    return plan


def test_shard_plan_c_abi_equals_numpy_formulation():
    from oracle import oracle_np as O
    from splicedice_amd import shard, synth
    cases = [synth.make_junctions(5000, 5, n_chrom=2), synth.make_junctions(3000, 6, n_chrom=1, gene_spacing=40, len_span=150000),
             synth.make_junctions(7, 7), synth.make_junctions(1, 8)]
    for cr, l, r, st in cases:
        _, row_ptr, col = O.cluster_csr(cr, l, r, st)
        for world in (1, 2, 3, 8, 16):
            for frac in (0.02, 0.0, 0.5):
                assert shard.shard_plan(row_ptr, col, world, frac) == _shard_plan_numpy(row_ptr, col, world, frac), (cr.size, world, frac)
    assert shard.shard_plan(np.zeros(1, np.int64), np.zeros(0, np.int32), 3) == [dict(own_lo=0, own_hi=0, ext_lo=0, ext_hi=0)] * 3


def test_junction_bed_writer_matches_reference_file(golden_dir, tmp_path):
    """sdice_write_junction_bed against a `_junctions.bed` the reference wrote (SPLICEDICE.py:316-321) and
    against the format string on awkward names"""
    from splicedice_amd import textio
    want = open(os.path.join(golden_dir, "quant_c1", "expected_default", "out_junctions.bed")).read()
    chrom_names, chrom, left, right, strand = [], [], [], [], []
    for ln in want.splitlines():
        c, l, r, _, _, s = ln.split("\t")
        if c not in chrom_names:
            chrom_names.append(c)
        chrom.append(chrom_names.index(c)); left.append(int(l)); right.append(int(r)); strand.append("+-".index(s))
    out = tmp_path / "j.bed"
    textio.write_junction_bed(out, chrom_names, chrom, left, right, strand)
    assert out.read_text() == want
    rng = np.random.default_rng(4)
    names = ["chr1", "HLA-A*01:01", "KI270728.1", "2"]
    n = 30000                                            # several worker threads
    chrom = rng.integers(0, 4, n); left = rng.integers(0, 2 ** 31 - 2, n); right = left + rng.integers(0, 2, n)
    strand = rng.integers(0, 2, n)
    textio.write_junction_bed(out, names, chrom, left, right, strand)
    want = "".join(f"{names[c]}\t{l}\t{r}\t{names[c]}:{l}-{r}:{'+-'[s]}\t0\t{'+-'[s]}\n"
                   for c, l, r, s in zip(chrom.tolist(), left.tolist(), right.tolist(), strand.tolist()))
    assert out.read_text() == want
    textio.write_junction_bed(out, names, [], [], [], [])
    assert out.read_text() == ""
    with pytest.raises(RuntimeError):
        textio.write_junction_bed(out, names, [4], [1], [2], [0])


def test_interval_overlaps_equals_the_reference_loop():
    from splicedice_amd import textio
    rng = np.random.default_rng(12)
    n_groups = 5
    sizes = [0, 1, 40, 300, 7]
    grp_ptr = np.concatenate([[0], np.cumsum(sizes)])
    lo = rng.integers(0, 10000, grp_ptr[-1]); hi = lo + rng.integers(0, 3000, grp_ptr[-1])
    n = 20000
    g = rng.integers(-1, n_groups, n); a = rng.integers(0, 13000, n); b = a + rng.integers(0, 500, n)
    ptr, idx = textio.interval_overlaps(g, a, b, grp_ptr, lo, hi)
    assert ptr[0] == 0 and ptr[-1] == idx.size
    for e in rng.integers(0, n, 400).tolist() + [0, n - 1]:
        want = [] if g[e] < 0 else [k for k in range(grp_ptr[g[e]], grp_ptr[g[e] + 1])
                                     if (a[e] >= lo[k] and a[e] <= hi[k]) or (b[e] >= lo[k] and b[e] <= hi[k])]
        assert idx[ptr[e]:ptr[e + 1]].tolist() == want
    ptr, idx = textio.interval_overlaps([], [], [], [0], [], [])
    assert ptr.tolist() == [0] and idx.size == 0


@pytest.mark.parametrize("gtf,table,known", [("anno.gtf", "expected_out_gtf.tsv", False),
                                              ("anno_hits.gtf", "expected_out_gtf_hits.tsv", True)])
def test_annotation_columns_against_reference_table(golden_dir, tmp_path, gtf, table, known):
    """gene / overlapping / transcript_id of the annotated compare_sample_sets table (compareSampleSets.py:238-264);
    the second fixture (tests/golden/make_golden_annot.py) has exons bordering tested events: a junction of two
    transcripts of one gene, one of two genes, genes on identical coordinates, the other strand"""
    from splicedice_amd import compare_sample_sets as css, textio
    d = os.path.join(golden_dir, "compare")
    want = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(d, table))][1:]
    names = [w[0] for w in want]
    sfx = css.annotation_suffixes(names, os.path.join(d, gtf))
    assert sfx == ["\t" + "\t".join(w[8:11]) for w in want]
    assert any(w[9] for w in want) and any(w[8] != "nan" for w in want) == known
    # and the suffix path of the column writer
    out = tmp_path / "t.tsv"
    x = np.arange(len(names), dtype=np.float64) / 7
    textio.write_columns(out, "event\tx\tgene\toverlapping\ttranscript_id\n", names, [x], ["repr"], suffixes=sfx)
    got = out.read_text().splitlines()
    assert got[1:] == [f"{nm}\t{str(np.float64(v))}{s}" for nm, v, s in zip(names, x, sfx)]


def test_rank_over_m_by_reciprocal(tmp_path):
    """bh_cols.hip raw_bits_inv(): the ecdf factor rank / m of statsmodels' fdr_bh by a reciprocal and one exact residual
    step instead of an IEEE division -- equal to the division for every rank of 321 column lengths up to 2^18
    (tests/host_c/rank_over_m.c, compiled here with gcc; 405 M cases with 3000 random lengths were run once by hand)"""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_c", "rank_over_m.c")
    exe = str(tmp_path / "rank_over_m")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    r = subprocess.run([exe, "300"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("bad 0"), r.stdout


def test_ps_of_key_formula_reproduces_the_table():
    """ranksum.hip ps_of_key(): q = k * 0.001f; q = fma(fma(-q, 1000, k), 0.001f, q) must equal float32(k / 1000.0)
    (the '.3f' text read back as float32, compareSampleSets.py:202) for every k = 0..1000 -- exact rational
    arithmetic with one correct rounding per float32 operation."""
    from fractions import Fraction

    def rn32(x):                                  # round-to-nearest-even of a Fraction to float32
        if x == 0:
            return np.float32(0)
        f = np.float32(float(x))
        best = None
        for c in (np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))):
            d = abs(Fraction(float(c)) - x)
            even = (int(np.float32(c).view(np.uint32)) & 1) == 0
            if best is None or d < best[0] or (d == best[0] and even):
                best = (d, c)
        return np.float32(best[1])

    def fr(v):
        return Fraction(float(v))
    c001, c1000 = np.float32(0.001), np.float32(1000.0)
    for k in range(1001):
        kf = np.float32(k)
        q = rn32(fr(kf) * fr(c001))
        r = rn32(-fr(q) * fr(c1000) + fr(kf))
        q = rn32(fr(r) * fr(c001) + fr(q))
        assert q == np.float32(np.float64(k) / 1000.0), k
