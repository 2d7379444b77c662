#!/bin/bash
# rocprofv3 summaries for one round (run on the GPU box through gpurun from the repo root):
#   tools/profile_round.sh r02            (every workload)
#   tools/profile_round.sh r02 compare    (one workload's trace + traffic passes only; summarise locally afterwards)
# kernel-trace/stats and the PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# counters are never combined with the trace domains).  The program itself follows `--` (python3 bench.py).
R=${1:-r02}
ONLY=${2:-}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ -n "$ONLY" ]; then
  WL=$ONLY
  STEPS=20; [ $WL = pairwise ] && STEPS=3; [ $WL = e2e ] && STEPS=5
  echo "trace $WL"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_trace -- python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-verify --no-also > $OUT/${WL}_bench.json 2> $OUT/${WL}_trace.err
  STEPS=3; [ $WL = pairwise ] && STEPS=1; [ $WL = e2e ] && STEPS=2
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "pmc $WL $C"
    rocprofv3 --pmc $C --output-format csv -d $OUT/${WL}_pmc_$C -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/${WL}_pmc_$C.err
  done
  exit 0
fi
for WL in quant compare pairwise e2e; do
  STEPS=20; [ $WL = pairwise ] && STEPS=3; [ $WL = e2e ] && STEPS=5
  echo "trace $WL"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_trace -- python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-verify --no-also > $OUT/${WL}_bench.json 2> $OUT/${WL}_trace.err
done
echo "trace quant 2M x 500"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quant2m500_trace -- python3 bench.py --workload quant --junctions 2000000 --samples 500 --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-also > $OUT/quant2m500_bench.json 2> $OUT/quant2m500_trace.err
for WL in quant compare pairwise; do
  STEPS=3; [ $WL = pairwise ] && STEPS=1
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "pmc $WL $C"
    rocprofv3 --pmc $C --output-format csv -d $OUT/${WL}_pmc_$C -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/${WL}_pmc_$C.err
  done
done
for C in FETCH_SIZE WRITE_SIZE; do
  echo "pmc quant2m500 $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/quant2m500_pmc_$C -- python3 bench.py --workload quant --junctions 2000000 --samples 500 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/quant2m500_pmc_$C.err
  echo "pmc e2e $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/e2e_pmc_$C -- python3 bench.py --workload e2e --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/e2e_pmc_$C.err
done
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU; do
  echo "pmc quant $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/quant_sq_$C -- python3 bench.py --workload quant --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/quant_sq_$C.err
done
for C in SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU; do
  echo "pmc pairwise $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/pairwise_sq_$C -- python3 bench.py --workload pairwise --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> $OUT/pairwise_sq_$C.err
done
python3 tools/summarise_profiles.py $OUT $R
