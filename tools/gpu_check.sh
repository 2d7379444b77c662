#!/bin/bash
# One gpurun call's worth of checking: the whole GPU suite, the smoke entry, the default bench line.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_check.sh'
set -o pipefail
mkdir -p gpurun_out/check
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/check/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/check/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/check/smoke.log 2>&1; rc=$?
tail -2 gpurun_out/check/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 700 python bench.py > gpurun_out/check/bench_default.json 2> gpurun_out/check/bench_default.err; rc=$?
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/check/bench_default.json"))
def show(k, r):
    ro = r["roofline"]
    print(f"{k}: {r['value']:.4g} {r['unit']}  {r['ms_per_step']:.4g} ms/step  {ro['kernel']} {ro['avg_kernel_ms']:.4g} ms  frac {ro['frac']:.3f}  verify {r.get('verify')}")
show("headline", d)
for k, r in d.get("also", {}).items():
    show(k, r)
PY
exit $rc
