#!/usr/bin/env python3
"""TEST / BENCH INFRASTRUCTURE ONLY -- the CPU baseline on ALL host cores.

The reference is single threaded; its restatement (oracle_np.py) is timed on one core by
bench.py.  SURVEY 8(d) also asks for a second figure: the same restatement junction-sharded over
every host core with multiprocessing.  This script is that figure.  bench.py starts it as a
child process (it never touches the GPU): every worker builds its own seeded junction shard of
the same layout, all workers meet at a barrier, then each runs the restatement on its shard;
the rate is total units / (last finish - first start).

    python oracle/cpu_baseline_mp.py --workload quant --units-per-core 60000 --samples 100
prints one JSON object on stdout.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _worker(idx, args, barrier, q):
    import numpy as np
    from oracle import oracle_np as O
    from splicedice_amd import synth
    m, s = args.units_per_core, args.samples
    seed = 1000 + idx
    if args.workload in ("quant", "e2e"):
        cr, l, r, st = synth.make_junctions(m, seed)
        counts = synth.make_counts(m, s, seed + 1)
        g1, g2 = np.arange(0, s // 2, dtype=np.int32), np.arange(s // 2, s, dtype=np.int32)
        barrier.wait()
        t0 = time.time()
        row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
        ps, _ = O.calculate_psi(counts, row_ptr, col)
        if args.workload == "e2e":
            res = O.compare_rows(O.quantize3_fast(ps), g1, g2)
            O.bh_fdr(res["p"][res["tested"].astype(bool)])
        units = m * s
    elif args.workload == "compare":
        ps = synth.make_ps_matrix(m, s, seed)
        g1, g2 = np.arange(0, s // 2, dtype=np.int32), np.arange(s // 2, s, dtype=np.int32)
        barrier.wait()
        t0 = time.time()
        res = O.compare_rows(ps, g1, g2)
        O.bh_fdr(res["p"][res["tested"].astype(bool)])
        units = m
    else:  # pairwise: m junctions x `samples` columns, all pairs
        cr, l, r, st = synth.make_junctions(max(m, 64), seed)
        counts_in = synth.make_counts(max(m, 64), s, seed + 1)
        row_of, row_ptr, col = O.cluster_csr(cr, l, r, st)
        counts = np.zeros_like(counts_in)
        counts[row_of] = counts_in
        _, excl = O.calculate_psi_vectorised(counts, row_ptr, col)
        barrier.wait()
        t0 = time.time()
        O.fisher_pairs(counts[:m], excl[:m])
        units = m * s * (s - 1) // 2
    q.put((units, t0, time.time()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["quant", "compare", "pairwise", "e2e"], required=True)
    ap.add_argument("--units-per-core", type=int, required=True, help="junctions (rows) per worker")
    ap.add_argument("--samples", type=int, required=True)
    ap.add_argument("--cores", type=int, default=0)
    args = ap.parse_args()
    cores = args.cores or os.cpu_count() or 1
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(cores), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(i, args, barrier, q)) for i in range(cores)]
    for p in procs:
        p.start()
    got = [q.get() for _ in procs]
    for p in procs:
        p.join()
    units = sum(g[0] for g in got)
    wall = max(g[2] for g in got) - min(g[1] for g in got)
    print(json.dumps({"value": units / wall, "cores": cores, "seconds": round(wall, 2),
                      "units_per_core": args.units_per_core, "samples": args.samples}))


if __name__ == "__main__":
    main()
