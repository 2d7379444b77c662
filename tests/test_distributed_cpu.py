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

    def ps(self, counts, row_ptr, col):
        return O.calculate_psi_vectorised(counts, row_ptr, col)[0]

    def quantize3(self, ps):
        return O.quantize3_fast(ps)

    def ranksum(self, ps, g1, g2):
        return O.compare_rows(ps, g1, g2)

    def bh(self, p):
        return O.bh_fdr(p)


def _problem():
    from splicedice_amd import synth
    n, s = 1200, 16
    cr, l, r, st = synth.make_junctions(n, 17, n_chrom=3)
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
        counts, row_ptr, col, g1, g2 = _problem()
        out = distributed.quant_compare_sharded(OracleEngine(), distributed.GlooComm(), counts, row_ptr, col, g1, g2)
        q.put((rank, {k: v for k, v in out.items() if k != "plan"}, out["plan"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2])
def test_sharded_pipeline_equals_single_process(world):
    import torch.multiprocessing as mp
    from splicedice_amd import distributed
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
    for rank, out, plan in results:
        assert len(plan) == world and plan[0]["own_hi"] > 0
        for k in ("tested", "p", "z", "corrected", "med1", "med2", "mean1", "mean2", "delta"):
            assert np.array_equal(out[k], single[k]), (rank, k)
    assert single["tested"].sum() > 500 and (single["corrected"][single["tested"] == 1] >= single["p"][single["tested"] == 1]).all()
