#!/bin/bash
# rocprofv3 summaries for one round (run on the GPU box through gpurun from the repo root):
#   tools/profile_round.sh r01
# kernel-trace/stats and the two PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# counters are never combined with the trace domains).
set -e
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
for WL in quant compare pairwise e2e; do
  STEPS=20; [ $WL = pairwise ] && STEPS=3; [ $WL = e2e ] && STEPS=5
  echo "trace $WL"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_trace -- python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-verify > $OUT/${WL}_bench.json 2> $OUT/${WL}_trace.err
done
for WL in quant compare pairwise; do
  STEPS=3; [ $WL = pairwise ] && STEPS=1
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "pmc $WL $C"
    rocprofv3 --pmc $C --output-format csv -d $OUT/${WL}_pmc_$C -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/${WL}_pmc_$C.err
  done
done
python3 tools/summarise_profiles.py $OUT $R
