set -x
mkdir -p gpurun_out/r2f
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -m gpu -k "ranksum or compare" > gpurun_out/r2f/pytest_rs.log 2>&1 || { tail -40 gpurun_out/r2f/pytest_rs.log; exit 1; }
tail -3 gpurun_out/r2f/pytest_rs.log
python bench.py --workload compare --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2f/bench_compare.json 2> gpurun_out/r2f/bench_compare.err; python -c "
import json; d=json.load(open('gpurun_out/r2f/bench_compare.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['verify'])"
