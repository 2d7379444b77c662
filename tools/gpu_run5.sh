set -x
mkdir -p gpurun_out/r2e
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r2e/pytest_dist.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/pytest_dist.log
tail -15 gpurun_out/r2e/pytest_dist.log
SDICE_BENCH_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2e/bench_n2_strong.json 2> gpurun_out/r2e/bench_n2_strong.err; echo "rc=$?"
cat gpurun_out/r2e/bench_n2_strong.json; tail -5 gpurun_out/r2e/bench_n2_strong.err
SDICE_BENCH_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --weak > gpurun_out/r2e/bench_n2_weak.json 2> gpurun_out/r2e/bench_n2_weak.err; echo "rc=$?"
cat gpurun_out/r2e/bench_n2_weak.json
