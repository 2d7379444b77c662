#!/bin/bash
# full GPU suite, then the pairwise bench line
set -o pipefail
mkdir -p gpurun_out/r2e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2e/gpu_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r2e/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --workload pairwise --steps 5 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/r2e/pairwise.json 2> gpurun_out/r2e/pairwise.err
rc=$?
tail -c 600 gpurun_out/r2e/pairwise.json | head -c 400; echo; python3 -c "
import json; d=json.load(open('gpurun_out/r2e/pairwise.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['verify'])"
exit $rc
