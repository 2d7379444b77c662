set -x
mkdir -p gpurun_out/r2b
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_gpu.log
tail -15 gpurun_out/r2b/pytest_gpu.log
timeout -k 10 600 python tools/bench_cli.py > gpurun_out/r2b/bench_cli.log 2>&1; tail -12 gpurun_out/r2b/bench_cli.log
