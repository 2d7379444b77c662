mkdir -p gpurun_out/r2g gpurun_out/prof_r02
timeout -k 5 120 tools/mb/microbench > gpurun_out/r2g/microbench.log 2>&1; tail -2 gpurun_out/r2g/microbench.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for C in SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/prof_r02/pairwise_sq_$C -- python3 bench.py --workload pairwise --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-also > /dev/null 2> gpurun_out/prof_r02/pairwise_sq_$C.err
  grep -h "fisher_pairs" gpurun_out/prof_r02/pairwise_sq_$C/*/*_counter_collection.csv | head -2
done
