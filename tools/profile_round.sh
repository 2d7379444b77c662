#!/bin/bash
# rocprofv3 summaries for one round (run on the GPU box through gpurun from the repo root):
#   tools/profile_round.sh r04            (every workload)
#   tools/profile_round.sh r04 compare    (one workload's trace + traffic passes only; summarise locally afterwards)
#   tools/profile_round.sh r04 full       (kernel traces of BASELINE configs 4 and 5 at FULL size on the one GPU)
# kernel-trace/stats and the PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# counters are never combined with the trace domains).  The program itself follows `--` (python3 bench.py).
# Workloads: quant = the headline (2M x 500), quantc2 = BASELINE config 2 (1M x 100), compare, pairwise, e2e.
R=${1:-r04}
ONLY=${2:-}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
COMMON="--no-cpu-baseline --no-verify --no-also"
args_of() {   # workload -> bench.py arguments (>= 20 profiled launches after >= 20 warm-ups for the quant lines)
  case $1 in
    quant)    echo "--workload quant --steps 25 --warmup 20" ;;
    quantc2)  echo "--workload quant --junctions 1000000 --samples 100 --steps 40 --warmup 20" ;;
    compare)  echo "--workload compare --steps 30 --warmup 10" ;;
    pairwise) echo "--workload pairwise --steps 3 --warmup 2" ;;
    e2e)      echo "--workload e2e --steps 8 --warmup 3" ;;
  esac
}
short_args_of() {   # fewer steps for the counter passes (one counter per run)
  case $1 in
    quant)    echo "--workload quant --steps 3 --warmup 2" ;;
    quantc2)  echo "--workload quant --junctions 1000000 --samples 100 --steps 3 --warmup 2" ;;
    compare)  echo "--workload compare --steps 3 --warmup 1" ;;
    pairwise) echo "--workload pairwise --steps 1 --warmup 1" ;;
    e2e)      echo "--workload e2e --steps 2 --warmup 1" ;;
  esac
}
WLS="quant quantc2 compare pairwise e2e"
[ -n "$ONLY" ] && WLS=$ONLY
if [ "$ONLY" = "full" ]; then   # BASELINE configs 4 and 5 at full size: kernel trace + stats only
  echo "trace pairwisefull"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pairwisefull_trace -- python3 bench.py --workload pairwise --junctions 200000 --steps 2 --warmup 1 $COMMON > $OUT/pairwisefull_bench.json 2> $OUT/pairwisefull_trace.err
  echo "trace e2efull"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2efull_trace -- python3 bench.py --workload e2e --junctions 5000000 --samples 1000 --steps 3 --warmup 1 $COMMON > $OUT/e2efull_bench.json 2> $OUT/e2efull_trace.err
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "pmc pairwisefull $C"
    rocprofv3 --pmc $C --output-format csv -d $OUT/pairwisefull_pmc_$C -- python3 bench.py --workload pairwise --junctions 200000 --steps 1 --warmup 1 $COMMON > /dev/null 2> $OUT/pairwisefull_pmc_$C.err
    echo "pmc e2efull $C"
    rocprofv3 --pmc $C --output-format csv -d $OUT/e2efull_pmc_$C -- python3 bench.py --workload e2e --junctions 5000000 --samples 1000 --steps 1 --warmup 1 $COMMON > /dev/null 2> $OUT/e2efull_pmc_$C.err
  done
  exit 0
fi
for WL in $WLS; do
  echo "trace $WL"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_trace -- python3 bench.py $(args_of $WL) $COMMON > $OUT/${WL}_bench.json 2> $OUT/${WL}_trace.err
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "pmc $WL $C"
    rocprofv3 --pmc $C --output-format csv -d $OUT/${WL}_pmc_$C -- python3 bench.py $(short_args_of $WL) $COMMON > /dev/null 2> $OUT/${WL}_pmc_$C.err
  done
done
[ -n "$ONLY" ] && exit 0
for WL in quant quantc2 compare e2e; do
  for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
    echo "pmc $WL $C"
    rocprofv3 --pmc $C --output-format csv -d $OUT/${WL}_sq_$C -- python3 bench.py $(short_args_of $WL) $COMMON > /dev/null 2> $OUT/${WL}_sq_$C.err
  done
done
for C in SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU; do
  echo "pmc pairwise $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/pairwise_sq_$C -- python3 bench.py $(short_args_of pairwise) $COMMON > /dev/null 2> $OUT/pairwise_sq_$C.err
done
python3 tools/summarise_profiles.py $OUT $R
