#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cluster or quant or pipeline" > gpurun_out/r2f/cluster_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r2f/cluster_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/dbg_cluster_once.py > gpurun_out/r2f/dbg2.log 2>&1; tail -14 gpurun_out/r2f/dbg2.log
timeout -k 10 600 python tools/ab_cluster.py 1000000 "" cluster.sample_sort=0 cluster.spb=16 cluster.spb=12 > gpurun_out/r2f/ab_cluster.log 2>&1; tail -4 gpurun_out/r2f/ab_cluster.log
timeout -k 10 600 python tools/ab_cluster.py 5000000 "" cluster.sample_sort=0 cluster.spb=16 > gpurun_out/r2f/ab_cluster5.log 2>&1; tail -3 gpurun_out/r2f/ab_cluster5.log
