#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fisher or pairwise or bh" > gpurun_out/r2g/fisher_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r2g/fisher_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_bh.py 25000 19900 "" > gpurun_out/r2g/ab_bh.log 2>&1; tail -2 gpurun_out/r2g/ab_bh.log
OUT=gpurun_out/prof_r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pairwise_pmc_$C
  rocprofv3 --pmc $C --output-format csv -d $OUT/pairwise_pmc_$C -- python3 bench.py --workload pairwise --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/pairwise_pmc_$C.err
done
echo pmc done
