#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2d
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fisher" > gpurun_out/r2d/fisher_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r2d/fisher_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/ab_fisher.py 25000 200 "" fisher.unroll=6 fisher.unroll=8 fisher.unroll=8,fisher.refill=16 fisher.unroll=6,fisher.refill=16 fisher.unroll=4,fisher.refill=16 > gpurun_out/r2d/ab_fisher.log 2>&1
tail -20 gpurun_out/r2d/ab_fisher.log
