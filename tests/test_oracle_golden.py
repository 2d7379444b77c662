"""CPU: pins the oracle (oracle/oracle_np.py) against the golden fixtures that were generated
by running the reference itself (tests/golden/make_golden.py) and against the scipy call
sites the reference uses.  No GPU, no HIP library."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_np as O


def _npz(golden_dir, tag):
    return np.load(os.path.join(golden_dir, "arrays", f"cluster_psi_{tag}.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c", "dense"])
def test_cluster_and_psi_match_reference(golden_dir, tag):
    z = _npz(golden_dir, tag)
    row_of, row_ptr, col = O.cluster_csr(z["chrom_rank"], z["left"], z["right"], z["strand"])
    assert np.array_equal(row_of, z["row_of"])
    assert np.array_equal(row_ptr, z["row_ptr"])
    assert np.array_equal(col, z["col"])                       # neighbour ORDER included
    psi, excl = O.calculate_psi(z["counts_rows"], row_ptr, col)
    assert psi.tobytes() == z["psi"].tobytes()                 # bit-exact, NaNs included
    psi2, excl2 = O.calculate_psi_vectorised(z["counts_rows"], row_ptr, col)
    assert psi2.tobytes() == z["psi"].tobytes() and np.array_equal(excl, excl2)


def test_quant_files_match_oracle(golden_dir):
    """the reference's own _allClusters / _inclusionCounts / _allPS files re-derived by the oracle"""
    base = os.path.join(golden_dir, "quant_c1", "expected_default", "out")
    names, counts = [], []
    with open(base + "_inclusionCounts.tsv") as fh:
        fh.readline()
        for line in fh:
            row = line.rstrip().split("\t")
            names.append(row[0])
            counts.append([int(x) for x in row[1:]])
    juncs = []
    for nm in names:
        c, coords, st = nm.split(":")
        l, r = coords.split("-")
        juncs.append((c, int(l), int(r), st))
    clusters = O.get_clusters(juncs)
    with open(base + "_allClusters.tsv") as fh:
        for line, j in zip(fh, sorted(clusters)):
            name, lst = line.rstrip("\n").split("\t")
            assert name == f"{j[0]}:{j[1]}-{j[2]}:{j[3]}"
            want = [f"{o[0]}:{o[1]}-{o[2]}:{o[3]}" for o in clusters[j]]
            assert (lst.split(",") if lst else []) == want
    index = {j: i for i, j in enumerate(sorted(clusters))}
    row_ptr = np.zeros(len(juncs) + 1, np.int64)
    col = []
    for r, j in enumerate(sorted(clusters)):
        col.extend(index[o] for o in clusters[j])
        row_ptr[r + 1] = len(col)
    psi, _ = O.calculate_psi(np.array(counts), row_ptr, np.array(col, np.int32))
    with open(base + "_allPS.tsv") as fh:
        fh.readline()
        for line, row in zip(fh, psi):
            assert line.rstrip("\n").split("\t")[1:] == [f"{x:.3f}" for x in row]


def test_fisher_restatement_vs_scipy_kat(golden_dir):
    kats = json.load(open(os.path.join(golden_dir, "kat_fisher.json")))
    for t, p in kats:
        q = O.fisher_exact_restated(*t)
        assert q == p or abs(q - p) <= 1e-9 * p, (t, p, q)


def test_ranksums_restatement_vs_scipy_kat(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "kat_ranksums.json"))):
        x, y = np.float32(c["x"]), np.float32(c["y"])
        z, p = O.ranksums_restated(x, y)
        assert z == c["z"] and abs(p - c["p"]) <= 1e-14 * c["p"]
        assert float(np.median(x)) == c["med1"] and float(np.mean(y)) == c["mean2"]


def test_compare_rows_vs_reference_output(golden_dir):
    d = os.path.join(golden_dir, "compare")
    rows, data = [], []
    with open(os.path.join(d, "in_allPS.tsv")) as fh:
        cols = fh.readline().strip().split("\t")[1:]
        for line in fh:
            row = line.strip().split("\t")
            rows.append(row[0])
            data.append(row[1:])
    matrix = np.array(data, dtype="float32")
    g1 = [ln.split()[0] for ln in open(os.path.join(d, "m1.tsv"))]
    g2 = [ln.split()[0] for ln in open(os.path.join(d, "m2.tsv"))]
    i1 = np.nonzero(np.isin(cols, g1))[0]
    i2 = np.nonzero(np.isin(cols, g2))[0]
    for use_scipy in (True, False):
        r = O.compare_rows(matrix, i1, i2, use_scipy=use_scipy)
        keep = np.flatnonzero(r["tested"])
        q = O.bh_fdr(r["p"][keep])
        with open(os.path.join(d, "expected_out.tsv")) as fh:
            fh.readline()
            lines = fh.readlines()
        assert len(lines) == keep.size
        for n, (line, ri) in enumerate(zip(lines, keep)):
            f = line.rstrip("\n").split("\t")
            assert f[0] == rows[ri]
            assert f[1] == str(r["mean1"][ri]) and f[2] == str(r["mean2"][ri])
            assert f[3] == str(r["med1"][ri]) and f[4] == str(r["med2"][ri]) and f[5] == str(r["delta"][ri])
            assert abs(float(f[6]) - r["p"][ri]) <= 1e-12 * float(f[6])
            assert abs(float(f[7]) - q[n]) <= 1e-12 * float(f[7])      # BH: parity unpinned (scipy shim)


@pytest.mark.parametrize("mode", ["none", "pairwise", "all"])
def test_pairwise_vs_reference_output(golden_dir, mode):
    d = os.path.join(golden_dir, "pairwise")
    events, counts = [], []
    with open(os.path.join(d, "in_inclusionCounts.tsv")) as fh:
        fh.readline()
        for line in fh:
            row = line.rstrip().split("\t")
            events.append(row[0])
            counts.append([int(x) for x in row[1:]])
    counts = np.array(counts)
    clusters = {}
    for line in open(os.path.join(d, "in_allClusters.tsv")):
        parts = line.rstrip().split()
        clusters[parts[0]] = parts[1].split(",") if len(parts) == 2 else []
    idx = {e: i for i, e in enumerate(events)}
    row_ptr = np.zeros(len(events) + 1, np.int64)
    col = []
    for n, e in enumerate(events):
        col.extend(idx[o] for o in clusters[e] if o in idx)
        row_ptr[n + 1] = len(col)
    excl = O.pairwise_exclusions(counts, row_ptr, np.array(col, np.int32))
    p = O.fisher_pairs(counts, excl, use_scipy=False)
    if mode == "pairwise":
        p = O.bh_columns(p)
    elif mode == "all":
        p = O.bh_fdr(p.ravel()).reshape(p.shape)
    with open(os.path.join(d, f"expected_{mode}.tsv")) as fh:
        fh.readline()
        for line, row in zip(fh, p):
            want = np.array([float(x) for x in line.rstrip("\n").split("\t")[1:]])
            np.testing.assert_allclose(row, want, rtol=1e-9, atol=0)


def test_quantize3_forms_agree():
    rng = np.random.default_rng(0)
    k = np.arange(0, 1001) / 1000.0
    vals = np.concatenate([k.astype(np.float32), (k + 0.0005).astype(np.float32), rng.random(5000).astype(np.float32),
                           np.float32([np.nan, 1e-8, 0.9995])])
    assert np.array_equal(O.quantize3(vals), O.quantize3_fast(vals), equal_nan=True)


def test_bh_matches_scipy():
    from scipy.stats import false_discovery_control
    rng = np.random.default_rng(1)
    p = rng.random(1000) ** 2
    p[:10] = p[10]
    np.testing.assert_allclose(O.bh_fdr(p), false_discovery_control(p, method="bh"), rtol=1e-12, atol=0)
    assert O.bh_fdr(np.zeros(0)).size == 0


def test_chi2_restatement_vs_scipy_and_reference_output(golden_dir):
    from scipy.stats import chi2_contingency
    rng = np.random.default_rng(3)
    for _ in range(300):
        t = rng.integers(1, 400, size=4)
        w = chi2_contingency(t.reshape(2, 2))[1]
        q = O.chi2_yates_restated(*t)
        assert abs(q - w) <= 1e-12 * max(w, 1e-300)
    with pytest.raises(ValueError):
        O.chi2_yates_restated(3, 4, 0, 0)
    d = os.path.join(golden_dir, "pairwise")
    events, counts = [], []
    with open(os.path.join(d, "in_inclusionCounts_pos.tsv")) as fh:
        fh.readline()
        for line in fh:
            row = line.rstrip().split("\t")
            events.append(row[0])
            counts.append([int(x) for x in row[1:]])
    counts = np.array(counts)
    clusters = {}
    for line in open(os.path.join(d, "in_allClusters.tsv")):
        parts = line.rstrip().split()
        clusters[parts[0]] = parts[1].split(",") if len(parts) == 2 else []
    idx = {e: i for i, e in enumerate(events)}
    row_ptr = np.zeros(len(events) + 1, np.int64)
    col = []
    for n, e in enumerate(events):
        col.extend(idx[o] for o in clusters[e] if o in idx)
        row_ptr[n + 1] = len(col)
    excl = O.pairwise_exclusions(counts, row_ptr, np.array(col, np.int32))
    p = O.chi2_pairs(counts, excl, use_scipy=False)
    with open(os.path.join(d, "expected_chi2_none.tsv")) as fh:
        fh.readline()
        for line, row in zip(fh, p):
            want = np.array([float(x) for x in line.rstrip("\n").split("\t")[1:]])
            np.testing.assert_allclose(row, want, rtol=1e-12, atol=0)


def test_similarity_restatement_vs_reference_output(golden_dir):
    """oracle similarity_scores + the host reading rules reproduce the reference's score files."""
    from splicedice_amd import similarity as sim
    s, c = os.path.join(golden_dir, "similarity"), os.path.join(golden_dir, "compare")
    for vs, allps, want in ((os.path.join(c, "expected_out.tsv"), os.path.join(c, "in_allPS.tsv"), "expected_scores.tsv"),
                            (os.path.join(s, "in_vs.tsv"), os.path.join(s, "in_allPS.tsv"), "expected_scores_handmade.tsv")):
        events = sim.significant_events(vs)
        with open(allps) as fh:
            samples = fh.readline().rstrip().split("\t")[1:]
            names, rows = [], []
            for line in fh:
                row = line.rstrip().split("\t")
                names.append(row[0])
                rows.append([float(x) for x in row[1:]])
        mid, sign = sim.row_parameters(names, events)
        scores, counts = O.similarity_scores(np.array(rows), mid, sign)
        lines = [f"{sm}\t{sc / ct:0.03f}\t{sc}\t{ct}\n"
                 for sc, sm, ct in sorted(zip(scores.tolist(), samples, counts.tolist()), reverse=True)]
        assert "".join(lines) == open(os.path.join(s, want)).read()


def test_find_outliers_restatement_vs_reference_output(golden_dir):
    from splicedice_amd import find_outliers as fo
    d = os.path.join(golden_dir, "outliers")
    samples = fo.first_fields(os.path.join(d, "samples.tsv"))
    null = fo.first_fields(os.path.join(d, "null.tsv"))
    for tag in ("f32", "f64"):
        z = np.load(os.path.join(d, f"matrix_{tag}.npz"))
        for name, grp, cut in (("self", samples, 3), ("null", null, 3)) + ((("cutoff2", samples, 2),) if tag == "f32" else ()):
            lines = O.find_outlier_lines(z["rows"], z["cols"], z["data"], samples, grp, cut)
            want = open(os.path.join(d, f"expected_{tag}_{name}.txt")).read()
            assert "".join(line + "\n" for line in lines) == want, (tag, name)


def test_write_ps_values_f64_vs_reference_output(golden_dir):
    """The float64 PS restatement (counts_to_ps.py:58-70) pinned by the reference's own output on a fractional count
    table (tests/golden/make_golden_fractional.py): same '0.3f' text for every cell."""
    d = os.path.join(golden_dir, "counts_to_ps_fractional")
    lines = open(os.path.join(d, "in_inclusionCounts.tsv")).read().splitlines()
    names = [ln.split("\t", 1)[0] for ln in lines[1:]]
    counts = np.array([ln.split("\t")[1:] for ln in lines[1:]], dtype=float)
    index = {nm: i for i, nm in enumerate(names)}
    clusters = {}
    for ln in open(os.path.join(golden_dir, "quant_c1", "expected_default", "out_allClusters.tsv")):
        j, ov = ln.rstrip("\n").split("\t")
        clusters[j] = [o for o in ov.split(",") if o]

    def key(nm):
        c, co, st = nm.split(":")
        a, b = co.split("-")
        return (c, int(a), int(b), st)
    order = sorted(clusters, key=key)
    row_ptr = np.zeros(len(order) + 1, dtype=np.int64)
    col = []
    for r, nm in enumerate(order):
        col.extend(index[o] for o in clusters[nm])
        row_ptr[r + 1] = len(col)
    table = counts[[index[nm] for nm in order]]
    remap = {index[nm]: r for r, nm in enumerate(order)}
    col = np.array([remap[c] for c in col], dtype=np.int32)
    ps = O.write_ps_values_f64(table, row_ptr, col)
    want = open(os.path.join(d, "expected_c", "out_allPS.tsv")).read().splitlines()
    assert len(want) == len(order) + 1
    for r, nm in enumerate(order):
        assert want[r + 1] == nm + "\t" + "\t".join(f"{x:0.3f}" for x in ps[r])


def test_exact_fisher_referee_vs_scipy_on_small_tables():
    """tools/exact_fisher.py (rational arithmetic; the referee for p-values near the underflow limit, where scipy is
    erratic) agrees with scipy where scipy is sound: small and medium tables, both sides of the mode, one-sided sums."""
    import importlib.util
    from scipy.stats import fisher_exact
    spec = importlib.util.spec_from_file_location(
        "exact_fisher", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "exact_fisher.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(12)
    tables = [(3, 1, 1, 3), (30, 12, 11, 35), (0, 5, 7, 2), (9, 0, 1, 8), (50, 50, 50, 50), (1, 200, 300, 2)]
    tables += [tuple(int(x) for x in rng.integers(1, 400, size=4)) for _ in range(40)]
    for a, b, c, d in tables:
        want = fisher_exact([[a, b], [c, d]])[1]
        got = mod.exact_two_sided(a, b, c, d)
        assert abs(got - want) <= 1e-10 * want, (a, b, c, d, got, want)
