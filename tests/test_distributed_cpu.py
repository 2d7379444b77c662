"""CPU, world_size 2 (gloo): the N>1 logic of the sharded quant -> compare pipeline
(shard plan, local CSRs, padded all-gather, global BH) gives exactly the single-process
result on every rank.  The compute engine here is an oracle-backed stand-in with the same
methods as the HIP Context -- this test is about the distribution logic; the same pipeline
runs on the real engine in tests/test_gpu_distributed.py."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle_np as O


class OracleEngine:
    """Test double with the engine.Context surface used by distributed.quant_compare_sharded."""

    def ps(self, counts, row_ptr, col, want_excl=False, want_ps=True):
        ps, excl = O.calculate_psi_vectorised(counts, row_ptr, col)
        return (ps, excl) if want_excl and want_ps else excl if want_excl else ps

    def fisher_pairs(self, incl, excl):
        return O.fisher_pairs(incl, excl)

    def chi2_pairs(self, incl, excl):
        """-> (p, number of tables with a zero expected frequency), the shape of engine.Context.chi2_pairs"""
        n, s = incl.shape
        pairs = O.pair_list(s)
        out, bad = np.ones((n, len(pairs))), 0
        for r in range(n):
            for q, (i, j) in enumerate(pairs):
                try:
                    out[r, q] = O.chi2_yates_restated(incl[r, i], incl[r, j], excl[r, i], excl[r, j])
                except ValueError:
                    bad += 1
        return out, bad

    def bh_columns(self, p):
        return O.bh_columns(p)

    def quantize3(self, ps):
        return O.quantize3_fast(ps)

    def ranksum(self, ps, g1, g2):
        return O.compare_rows(ps, g1, g2)

    def bh(self, p):
        return O.bh_fdr(p)

    def cluster(self, cr, l, r, st):
        return O.cluster_csr(cr, l, r, st)


def _junction_problem(kw=None):
    """the same junction set as _problem, in output row order, with its counts: what a rank that clusters ITS OWN
    range works from (no global CSR anywhere)"""
    from splicedice_amd import shard, synth
    n, s = 1200, 16
    cr, l, r, st = synth.make_junctions(n, 17, **(kw or dict(n_chrom=3)))
    o = shard.junction_order(cr, l, r, st)
    counts, _, _, g1, g2 = _problem(kw)
    return tuple(x[o] for x in (cr, l, r, st)), counts, g1, g2


def _problem(kw=None):
    from splicedice_amd import synth
    n, s = 1200, 16
    cr, l, r, st = synth.make_junctions(n, 17, **(kw or dict(n_chrom=3)))
    row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
    counts_in = synth.make_counts(n, s, 18, mean=12)
    counts = np.zeros_like(counts_in)
    counts[row_of] = counts_in
    counts[::37] = 0                                     # some untested / NaN rows
    g1, g2 = np.arange(0, 8, dtype=np.int32), np.arange(8, 16, dtype=np.int32)
    return counts, row_ptr, col, g1, g2


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from splicedice_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from splicedice_amd import shard
        counts, row_ptr, col, g1, g2 = _problem()
        comm = distributed.GlooComm()
        part = shard.shard_plan(row_ptr, col, world)[rank]
        mine = counts[part["ext_lo"]:part["ext_hi"]].copy()      # a rank is handed ITS rows only
        del counts
        out = distributed.quant_compare_sharded(OracleEngine(), comm, mine, row_ptr, col, g1, g2)
        # the same run with every rank clustering ITS OWN range from the junction coordinates (no global CSR): dense
        # junctions so that the cuts are not clean and the shards carry halos
        res2 = {}
        for tag, kw in (("genes", None), ("dense", dict(n_chrom=2, gene_spacing=300, len_span=60000))):
            junc, counts2, g1, g2 = _junction_problem(kw)
            plan2 = shard.shard_plan_junctions(*junc, world)
            a, b = plan2[rank]["ext_lo"], plan2[rank]["ext_hi"]
            o2 = distributed.quant_compare_sharded(OracleEngine(), comm, counts2[a:b].copy(), None, None, g1, g2, plan=plan2,
                                                   junctions_ext=tuple(x[a:b] for x in junc))
            pw = distributed.pairwise_sharded(OracleEngine(), comm, counts2[a:b, :5].copy(), None, None, "pairwise", plan=plan2,
                                              junctions_ext=tuple(x[a:b] for x in junc))
            res2[tag] = ({k: v for k, v in o2.items() if k != "plan"}, plan2, pw["own"], pw["p"])
        q.put((rank, {k: v for k, v in out.items() if k != "plan"}, out["plan"], res2))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2])
def test_sharded_pipeline_equals_single_process(world):
    import torch.multiprocessing as mp
    from splicedice_amd import distributed, shard
    counts, row_ptr, col, g1, g2 = _problem()
    single = distributed.quant_compare_sharded(OracleEngine(), distributed.SingleComm(), counts, row_ptr, col, g1, g2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    keys = ("tested", "p", "z", "corrected", "med1", "med2", "mean1", "mean2", "delta")
    singles = {}
    for tag, kw in (("genes", None), ("dense", dict(n_chrom=2, gene_spacing=300, len_span=60000))):
        c2, rp2, col2, g1b, g2b = _problem(kw)
        one = distributed.quant_compare_sharded(OracleEngine(), distributed.SingleComm(), c2, rp2, col2, g1b, g2b)
        _, excl2 = O.calculate_psi_vectorised(c2[:, :5], rp2, col2)
        singles[tag] = (one, shard.shard_plan(rp2, col2, world), O.bh_columns(O.fisher_pairs(c2[:, :5], excl2)))
    assert any(p["ext_lo"] < p["own_lo"] or p["ext_hi"] > p["own_hi"] for p in singles["dense"][1])      # halos are exercised
    for rank, out, plan, res2 in results:
        assert len(plan) == world and plan[0]["own_hi"] > 0
        for k in keys:
            assert np.array_equal(out[k], single[k]), (rank, k)
        for tag, (o2, plan2, own, pw) in res2.items():
            one, plan_csr, pw_want = singles[tag]
            assert plan2 == plan_csr, (rank, tag)                    # the coordinate plan IS the CSR plan
            for k in keys:
                assert np.array_equal(o2[k], one[k]), (rank, tag, k)  # own-range clustering == replicated clustering
            assert np.array_equal(pw, pw_want[own[0]:own[1]]), (rank, tag)
    assert single["tested"].sum() > 500 and (single["corrected"][single["tested"] == 1] >= single["p"][single["tested"] == 1]).all()


# ---------------------------------------------------------------- pairwise: column BH across ranks
def _pairwise_problem():
    from splicedice_amd import synth
    n, s = 260, 6
    cr, l, r, st = synth.make_junctions(n, 19, n_chrom=2)
    row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
    counts_in = synth.make_counts(n, s, 20, mean=15)
    counts = np.zeros_like(counts_in)
    counts[row_of] = counts_in
    return counts, row_ptr, col


def _chi2_problem():
    """40 rows in overlapping pairs (2i <-> 2i + 1), 5 samples, counts >= 1: every 2x2 table has positive margins"""
    n, s = 40, 5
    rng = np.random.default_rng(77)
    counts = rng.integers(1, 60, size=(n, s)).astype(np.int32)
    row_ptr = np.arange(n + 1, dtype=np.int64)
    col = (np.arange(n) ^ 1).astype(np.int32)
    return counts, row_ptr, col


def _pairwise_worker(rank, world, port, q):
    import torch.distributed as dist
    from splicedice_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from splicedice_amd import shard
        counts, row_ptr, col = _pairwise_problem()
        part = shard.shard_plan(row_ptr, col, world)[rank]
        mine = counts[part["ext_lo"]:part["ext_hi"]].copy()      # a rank is handed ITS rows only
        del counts
        res = {}
        for mode in ("pairwise", "none", "all"):
            out = distributed.pairwise_sharded(OracleEngine(), distributed.GlooComm(), mine, row_ptr, col, mode)
            res[mode] = (out["own"], out["p"])
        # --chi2 on shards of its own small problem (rows in overlapping pairs: no empty row or column anywhere) ...
        c2, rp2, col2 = _chi2_problem()
        part2 = shard.shard_plan(rp2, col2, world)[rank]
        mine2 = c2[part2["ext_lo"]:part2["ext_hi"]].copy()
        out = distributed.pairwise_sharded(OracleEngine(), distributed.GlooComm(), mine2, rp2, col2, "pairwise", test="chi2")
        res["chi2"] = (out["own"], out["p"])
        # ... and the abort: ONE table with a zero expected frequency, in the LAST rank's rows, stops every rank
        if rank == world - 1:
            mine2[part2["own_hi"] - part2["ext_lo"] - 1, :2] = 0
            mine2[part2["own_hi"] - part2["ext_lo"] - 2, :2] = 0
        try:
            distributed.pairwise_sharded(OracleEngine(), distributed.GlooComm(), mine2, rp2, col2, "none", test="chi2")
            res["chi2_abort"] = "no error"
        except ValueError as e:
            res["chi2_abort"] = str(e)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_pairwise_column_bh_equals_single_process():
    """rows -> columns all-to-all, BH on complete pair columns, all-to-all back == the unsharded
    reference semantics (pairwise_fisher.py:164-191)."""
    import torch.multiprocessing as mp
    from splicedice_amd import distributed
    world = 2
    counts, row_ptr, col = _pairwise_problem()
    _, excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    raw = O.fisher_pairs(counts, excl)
    want = {"none": raw, "pairwise": O.bh_columns(raw), "all": O.bh_fdr(raw.reshape(-1)).reshape(raw.shape)}
    single = distributed.pairwise_sharded(OracleEngine(), distributed.SingleComm(), counts, row_ptr, col, "pairwise")
    assert single["own"] == (0, counts.shape[0]) and np.array_equal(single["p"], want["pairwise"])
    single_all = distributed.pairwise_sharded(OracleEngine(), distributed.SingleComm(), counts, row_ptr, col, "all")
    assert np.array_equal(single_all["p"], want["all"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pairwise_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c2, rp2, col2 = _chi2_problem()
    _, excl2 = O.calculate_psi_vectorised(c2, rp2, col2)
    p2, bad2 = OracleEngine().chi2_pairs(c2, excl2)
    assert bad2 == 0
    want["chi2"] = O.bh_columns(p2)
    covered = 0
    for rank, res in results:
        for mode in ("pairwise", "none", "all", "chi2"):
            (lo, hi), got = res[mode]
            assert np.array_equal(got, want[mode][lo:hi]), (rank, mode)
        assert res["chi2_abort"].startswith(distributed.CHI2_ZERO_MSG), (rank, res["chi2_abort"])
        covered += res["none"][0][1] - res["none"][0][0]
    assert covered == counts.shape[0]
    assert distributed.pair_column_ranges(15, 4) == [(0, 3), (3, 7), (7, 11), (11, 15)]


# ---------------------------------------------------------------- sub-commands under a 2-rank launcher
def _cli_worker(rank, world, port, which, outdir, golden):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import argparse
    from splicedice_amd import compare_sample_sets, pairwise
    eng = OracleEngine()
    if which == "compare":
        d = os.path.join(golden, "compare")
        ns = argparse.Namespace(psiSPLICEDICE=os.path.join(d, "in_allPS.tsv"), manifest1=os.path.join(d, "m1.tsv"),
                                manifest2=os.path.join(d, "m2.tsv"), annotation="", outputFile=os.path.join(outdir, "out.tsv"))
        compare_sample_sets.run_with(ns, ctx=eng)
    else:
        d = os.path.join(golden, "pairwise")
        for mode in ("pairwise", "all", "none"):
            ns = argparse.Namespace(inclusionSPLICEDICE=os.path.join(d, "in_inclusionCounts.tsv"),
                                    clusters=os.path.join(d, "in_allClusters.tsv"), chi2=False,
                                    multiple_test_correction=mode, filter_list=None, output=os.path.join(outdir, f"{mode}.tsv"))
            pairwise.run_with(ns, ctx=eng)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("which", ["compare", "pairwise"])
def test_subcommands_under_two_rank_launcher(which, tmp_path, golden_dir):
    """`python -m torch.distributed.run --nproc-per-node 2 -m splicedice_amd <subcommand>`: both ranks run the
    sub-command on their rows, ONE set of output files appears, equal to the reference-generated goldens
    (the compute engine is the oracle-backed double; the real engine takes the same path with RcclComm)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    procs = [mpctx.Process(target=_cli_worker, args=(r, 2, port, which, str(tmp_path), golden_dir)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    def cells(path):
        with open(path) as fh:
            return [line.rstrip("\n").split("\t") for line in fh]
    if which == "compare":
        got, want = cells(tmp_path / "out.tsv"), cells(os.path.join(golden_dir, "compare", "expected_out.tsv"))
        assert [g[0] for g in got] == [w[0] for w in want] and got[0] == want[0]
        for g, w in zip(got[1:], want[1:]):
            assert g[1:6] == w[1:6]                                              # float32 cells: string-identical
            np.testing.assert_allclose([float(x) for x in g[6:]], [float(x) for x in w[6:]], rtol=1e-9)
    else:
        for mode in ("pairwise", "all", "none"):
            got, want = cells(tmp_path / f"{mode}.tsv"), cells(os.path.join(golden_dir, "pairwise", f"expected_{mode}.tsv"))
            assert [g[0] for g in got] == [w[0] for w in want] and got[0] == want[0]
            for g, w in zip(got[1:], want[1:]):
                np.testing.assert_allclose([float(x) for x in g[1:]], [float(x) for x in w[1:]], rtol=1e-9)
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
