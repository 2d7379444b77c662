#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2i
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2i/gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r2i/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/ab_bh.py 25000 19900 "" bh.spb=16 bh.mean=176,bh.spb=16 bh.mean=144 bh.mean=192,bh.spb=24 > gpurun_out/r2i/ab_bh.log 2>&1; tail -5 gpurun_out/r2i/ab_bh.log
timeout -k 10 400 python tools/ab_bh.py 200000 2488 "" bh.spb=16 > gpurun_out/r2i/ab_bh200k.log 2>&1; tail -2 gpurun_out/r2i/ab_bh200k.log
