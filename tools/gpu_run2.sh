set -x
mkdir -p gpurun_out/r2b
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_gpu.log
tail -6 gpurun_out/r2b/pytest_gpu.log
python bench.py > gpurun_out/r2b/bench_default.json 2> gpurun_out/r2b/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2b/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['verify'], d['cpu_baseline']['value'] if d['cpu_baseline'] else None)
for k,v in d.get('also',{}).items():
    print(k, v.get('error') or (v['value'], round(v['ms_per_step'],3), round(v['roofline']['frac'],3), v['roofline']['kernel'], v['verify'], v.get('cpu_baseline',{}).get('value'), v['wall_seconds']))
PY
