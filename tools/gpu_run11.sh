#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cluster or quant or pipeline" > gpurun_out/r2f/cluster_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r2f/cluster_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ablate_cluster.py 1000000 0 -9999 0 -9999 > gpurun_out/r2f/ablate.log 2>&1
tail -5 gpurun_out/r2f/ablate.log
timeout -k 10 300 python tools/ablate_cluster.py 5000000 0 -9999 > gpurun_out/r2f/ablate5m.log 2>&1
tail -3 gpurun_out/r2f/ablate5m.log
