#!/bin/bash
# rocprofv3 summaries for one round (run on the GPU box through gpurun from the repo root):
#   tools/profile_round.sh r01
# kernel-trace/stats and the two PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
for WL in quant compare pairwise; do
  STEPS=20; [ $WL = pairwise ] && STEPS=3
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_trace -- python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-verify > $OUT/${WL}_bench.json 2> $OUT/${WL}_trace.err
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/quant_pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/quant_pmc_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_write.err
python3 tools/summarise_profiles.py $OUT $R
