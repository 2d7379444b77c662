#!/bin/bash
# BH sample-sort path: parity, then the config-4 shard bench line
set -o pipefail
mkdir -p gpurun_out/r2d
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bh" > gpurun_out/r2d/bh_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r2d/bh_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --workload pairwise --steps 5 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/r2d/pairwise.json 2> gpurun_out/r2d/pairwise.err
rc=$?
tail -c 1500 gpurun_out/r2d/pairwise.json; tail -3 gpurun_out/r2d/pairwise.err
exit $rc
