#!/bin/bash
# SQ counters of the compare (config 3) step, one counter per run
OUT=gpurun_out/prof_r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/compare_sq_$C -- python3 bench.py --workload compare --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/compare_sq_$C.err
  f=$(ls -t $OUT/compare_sq_$C/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" $C <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and "ranksum_pairq" in r["Kernel_Name"]:
        agg["pairq"].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[2], k, "%.4g per launch" % (sum(v) / len(v)), flush=True)
PY
done
