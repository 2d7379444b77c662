#!/bin/bash
# SQ counters of the e2e (config-5 shard) step's count kernel, one counter per run
OUT=gpurun_out/prof_r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/e2e_sq_$C -- python3 bench.py --workload e2e --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/e2e_sq_$C.err
  f=$(ls -t $OUT/e2e_sq_$C/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" $C <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and ("ranksum_count" in r["Kernel_Name"] or "ps_tile" in r["Kernel_Name"]):
        agg["count" if "ranksum_count" in r["Kernel_Name"] else "ps"].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[2], k, "%.4g per launch" % (sum(v) / len(v)), flush=True)
PY
done
