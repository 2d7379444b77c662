set -x
START=$(date +%s)
mkdir -p gpurun_out/r2d
python bench.py > gpurun_out/r2d/bench_default.json 2> gpurun_out/r2d/bench_default.err; echo "rc=$? wall=$(( $(date +%s) - START ))s"
tail -5 gpurun_out/r2d/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2d/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['verify'], d['cpu_baseline']['value'] if d['cpu_baseline'] else None)
for k,v in d.get('also',{}).items():
    print(k, v.get('error') or (v['value'], round(v['ms_per_step'],3), round(v['roofline']['frac'],3), v['roofline']['kernel'], v['verify'], v.get('cpu_baseline',{}).get('value'), v['wall_seconds'], v['gen_seconds']))
PY
