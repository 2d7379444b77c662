"""GPU: the sharded quant -> compare pipeline on the real engine (world 1 on the one-GPU box),
against the oracle, plus the library's RCCL communicator at world size 1."""
import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _problem():
    from splicedice_amd import synth
    n, s = 5000, 24
    cr, l, r, st = synth.make_junctions(n, 27, n_chrom=3)
    row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
    counts_in = synth.make_counts(n, s, 28, mean=12)
    counts = np.zeros_like(counts_in)
    counts[row_of] = counts_in
    counts[::41] = 0
    return counts, row_ptr, col, np.arange(0, 12, dtype=np.int32), np.arange(12, 24, dtype=np.int32)


def test_sharded_pipeline_on_engine_matches_oracle(ctx):
    from splicedice_amd import distributed, shard
    counts, row_ptr, col, g1, g2 = _problem()
    out = distributed.quant_compare_sharded(ctx, distributed.SingleComm(), counts, row_ptr, col, g1, g2)
    ps = O.quantize3_fast(O.calculate_psi_vectorised(counts, row_ptr, col)[0])
    want = O.compare_rows(ps, g1, g2)
    t = want["tested"].astype(bool)
    assert np.array_equal(out["tested"], want["tested"]) and np.array_equal(out["z"][t], want["z"][t])
    for k in ("med1", "med2", "mean1", "mean2", "delta"):
        assert np.array_equal(out[k][t], want[k][t])
    np.testing.assert_allclose(out["p"][t], want["p"][t], rtol=1e-9, atol=0)
    np.testing.assert_allclose(out["corrected"][t], O.bh_fdr(want["p"][t]), rtol=1e-9, atol=0)
    # every shard of a 4-way plan, computed one after the other on this GPU, reproduces its rows
    plan = shard.shard_plan(row_ptr, col, 4)
    for part in plan:
        rp, cl = shard.local_csr(row_ptr, col, part)
        loc = ctx.ps(counts[part["ext_lo"]:part["ext_hi"]], rp, cl)
        a, b = part["own_lo"] - part["ext_lo"], part["own_hi"] - part["ext_lo"]
        full = O.calculate_psi_vectorised(counts, row_ptr, col)[0]
        assert np.array_equal(loc[a:b], full[part["own_lo"]:part["own_hi"]], equal_nan=True)
        # and the device-resident shard object (PS -> quantise -> rank-sum into ONE packed block without leaving HBM)
        sh = distributed.CompareShard(ctx, distributed.SingleComm(), counts.shape[0], counts.shape[1], [part], g1, g2)
        try:
            sh.load(np.ascontiguousarray(counts[part["ext_lo"]:part["ext_hi"]]), rp, cl)
            sh.step()
            ctx.sync()
            st = sh.result()
        finally:
            sh.free()
        assert all(v.shape[0] == b - a for v in st.values())
        sl = slice(part["own_lo"], part["own_hi"])
        tt = want["tested"][sl].astype(bool)
        assert np.array_equal(st["tested"], want["tested"][sl]) and np.array_equal(st["z"][tt], want["z"][sl][tt])
        assert np.array_equal(st["med1"][tt], want["med1"][sl][tt]) and np.array_equal(st["mean2"][tt], want["mean2"][sl][tt])


@pytest.mark.parametrize("kw", [dict(n_chrom=3), dict(n_chrom=2, gene_spacing=300, len_span=60000)])
def test_own_range_clustering_equals_replicated_clustering(ctx, kw):
    """every shard of a 3-way plan made from the junction COORDINATES (no CSR): the device pipelines cluster rows
    [ext_lo, ext_hi) themselves (sdice_cluster_dev inside the step) and reproduce, bit for bit, what the same pipelines
    give on the local CSR cut out of a clustering of the whole set -- quant -> compare and pairwise (dense junctions:
    shards with halos)"""
    from splicedice_amd import distributed, shard, synth
    n, s = 6000, 24
    cr, l, r, st = synth.make_junctions(n, 91, **kw)
    o = shard.junction_order(cr, l, r, st)
    junc = tuple(x[o] for x in (cr, l, r, st))
    row_of, row_ptr, col = ctx.cluster(*junc)
    assert np.array_equal(row_of, np.arange(n))
    counts = synth.make_counts(n, s, 92, mean=12)
    g1, g2 = np.arange(0, 12, dtype=np.int32), np.arange(12, 24, dtype=np.int32)
    plan = shard.shard_plan_junctions(*junc, 3)
    assert plan == shard.shard_plan(row_ptr, col, 3)
    halos = 0
    for part in plan:
        a, b = part["ext_lo"], part["ext_hi"]
        halos += (a < part["own_lo"]) + (b > part["own_hi"])
        ext, jext = np.ascontiguousarray(counts[a:b]), tuple(x[a:b] for x in junc)
        rp, cl = shard.local_csr(row_ptr, col, part)
        got = {}
        for mode in ("csr", "junctions"):
            sh = distributed.CompareShard(ctx, distributed.SingleComm(), n, s, [part], g1, g2)
            try:
                sh.load(ext, rp, cl) if mode == "csr" else sh.load(ext, junctions=jext)
                sh.step()
                sh.step()                                           # (a second step re-clusters into the same buffers)
                ctx.sync()
                got[mode] = sh.result()
            finally:
                sh.free()
        for k, v in got["csr"].items():
            assert np.array_equal(v, got["junctions"][k], equal_nan=True), k
        pw = {}
        for mode in ("csr", "junctions"):
            # (correction "none": the raw p-values of the shard's rows -- column BH needs every rank's rows)
            sh = distributed.PairwiseShard(ctx, distributed.SingleComm(), n, 8, [part], "none", "fisher")
            try:
                e8 = np.ascontiguousarray(ext[:, :8])
                sh.load(e8, rp, cl) if mode == "csr" else sh.load(e8, junctions=jext)
                sh.step()
                pw[mode] = sh.result()
            finally:
                sh.free()
        assert np.array_equal(pw["csr"], pw["junctions"])
    assert ("gene_spacing" not in kw) or halos > 0


def test_rccl_world1_allgather():
    """own context: the communicator lives and dies with it"""
    from splicedice_amd.engine import Context
    from splicedice_amd import distributed
    with Context(0) as c:
        comm = distributed.RcclComm(c, 0, 1, lambda b, n: b)
        table = np.arange(40, dtype=np.float64).reshape(5, 8)
        assert np.array_equal(comm.allgather(table), table)
        d = c.to_device(table)
        assert np.array_equal(comm.allgather(d).to_host(), table)


def test_sharded_pairwise_on_engine(ctx):
    """pairwise_sharded on the HIP engine (world 1): exclusion sums + Fisher + column BH through the
    same code path the multi-rank run takes, incl. the library's all-to-all (RcclComm, world 1)."""
    from splicedice_amd import distributed, synth
    from splicedice_amd.engine import Context
    n, s = 700, 9
    cr, l, r, st = synth.make_junctions(n, 33, n_chrom=2)
    row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
    counts_in = synth.make_counts(n, s, 34, mean=20)
    counts = np.zeros_like(counts_in)
    counts[row_of] = counts_in
    _, excl = O.calculate_psi_vectorised(counts, row_ptr, col)
    raw = O.fisher_pairs(counts[:60], excl[:60])
    out = distributed.pairwise_sharded(ctx, distributed.SingleComm(), counts, row_ptr, col, "none")
    np.testing.assert_allclose(out["p"][:60], raw, rtol=1e-9, atol=0)
    full_raw = out["p"]
    out = distributed.pairwise_sharded(ctx, distributed.SingleComm(), counts, row_ptr, col, "pairwise")
    np.testing.assert_allclose(out["p"], O.bh_columns(full_raw), rtol=1e-12, atol=0)
    with Context(0) as c:
        comm = distributed.RcclComm(c, 0, 1, lambda b, n_: b)
        out2 = distributed.pairwise_sharded(c, comm, counts, row_ptr, col, "pairwise")
        assert np.array_equal(out2["p"], out["p"])
        blocks = np.arange(12, dtype=np.float64).reshape(1, 3, 4)
        assert np.array_equal(comm.alltoall(blocks), blocks)
        # the way home in column groups on the context's second stream (sdice_comm_fork / _join), pitched column BH per group
        for groups in (3, 7):
            outg = distributed.pairwise_sharded(c, comm, counts, row_ptr, col, "pairwise", overlap_groups=groups)
            assert np.array_equal(outg["p"], out["p"]), groups
        outg = distributed.pairwise_sharded(ctx, distributed.SingleComm(), counts, row_ptr, col, "pairwise", overlap_groups=4)
        assert np.array_equal(outg["p"], out["p"])
        out3 = distributed.pairwise_sharded(c, comm, counts, row_ptr, col, "all")
        np.testing.assert_allclose(out3["p"], O.bh_fdr(full_raw.reshape(-1)).reshape(full_raw.shape), rtol=1e-12, atol=0)
