set -x
mkdir -p gpurun_out/r2b
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=8 > gpurun_out/r2b/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_full.log
tail -30 gpurun_out/r2b/pytest_full.log
