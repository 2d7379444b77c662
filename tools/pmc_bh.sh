#!/bin/bash
# counter passes over the column-BH kernels (gpurun, repo root): tools/pmc_bh.sh TAG [params]
TAG=${1:-bh}; PARAMS=${2:-}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$OUT/$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $D -- python3 tools/run_bh_once.py 25000 19900 1 "$PARAMS" > $D.log 2>&1 || echo "pass $C failed"
done
python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(dict)
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    # rows are per dispatch (one per launch when summed over the dimensions): average per launch = total / launches
    launches = collections.Counter()
    for r in csv.DictReader(open(f)):
        launches[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["Counter_Name"], r["Dispatch_Id"])] += 1
    per = collections.Counter()
    for (k, c, d) in launches: per[(k, c)] += 1
    for (k, c), (v, _) in acc.items():
        out[k][c] = v / max(per[(k, c)], 1)
with open("$OUT/summary.csv", "w") as fh:
    names = sorted({c for v in out.values() for c in v})
    fh.write("kernel," + ",".join(names) + "\n")
    for k, v in sorted(out.items()):
        if "bhs" in k or "transpose" in k:
            fh.write(k + "," + ",".join(f"{v.get(c, float('nan')):.4g}" for c in names) + "\n")
print(open("$OUT/summary.csv").read())
PY
