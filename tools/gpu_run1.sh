set -x
mkdir -p gpurun_out/r2a
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cluster" > gpurun_out/r2a/pytest_cluster.log 2>&1 || { tail -30 gpurun_out/r2a/pytest_cluster.log; exit 1; }
tail -3 gpurun_out/r2a/pytest_cluster.log
timeout -k 10 300 python tools/ablate_cluster.py 1000000 0 64 0 > gpurun_out/r2a/ablate_1m.log 2>&1
cat gpurun_out/r2a/ablate_1m.log
