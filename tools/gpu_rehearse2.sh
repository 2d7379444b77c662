#!/bin/bash
# 2-rank rehearsal of bench.py's N>1 control flow on ONE GPU (RCCL refuses the duplicate device; the line says so)
mkdir -p gpurun_out/r2k
export SDICE_BENCH_DEVICE=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2k/n2_default.json 2> gpurun_out/r2k/n2_default.err
echo rc=$?; tail -c 900 gpurun_out/r2k/n2_default.json
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 5 --warmup 2 --strong > gpurun_out/r2k/n2_strong.json 2> gpurun_out/r2k/n2_strong.err
echo rc=$?; python3 -c "
import json
for f in ('n2_default','n2_strong'):
    d=json.load(open('gpurun_out/r2k/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['scaling'], d['config']['workload'][:80], d.get('ps_allgather'), d['verify'])"

# the product's sharded pipelines (pairwise: all-to-all + column BH; e2e: one packed all-gather + BH), small sizes
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --steps 3 --warmup 1 --workload pairwise --junctions 2000 --samples 40 > gpurun_out/r2k/n2_pairwise.json 2> gpurun_out/r2k/n2_pairwise.err
echo rc=$?; tail -c 1500 gpurun_out/r2k/n2_pairwise.json; tail -3 gpurun_out/r2k/n2_pairwise.err
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29536 bench.py --gpus 2 --steps 3 --warmup 1 --workload e2e --junctions 20000 --samples 100 > gpurun_out/r2k/n2_e2e.json 2> gpurun_out/r2k/n2_e2e.err
echo rc=$?; tail -c 1500 gpurun_out/r2k/n2_e2e.json; tail -3 gpurun_out/r2k/n2_e2e.err
