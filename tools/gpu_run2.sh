set -x
mkdir -p gpurun_out/r2b
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_gpu.log
tail -5 gpurun_out/r2b/pytest_gpu.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2b/bench_quant.json 2> gpurun_out/r2b/bench_quant.err; cat gpurun_out/r2b/bench_quant.json
python bench.py --workload e2e --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_e2e.json 2> gpurun_out/r2b/bench_e2e.err; cat gpurun_out/r2b/bench_e2e.json
