#!/bin/bash
# kernel-level breakdown of the pairwise (config-4 shard) step
set -o pipefail
out=$PWD/gpurun_out/r2d/prof_pw
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pw -- python3 $GRAFT_REPO_ROOT/bench.py --workload pairwise --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-verify > $out/bench.json 2> $out/bench.err
rc=$?
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-5 "$f" | cut -c1-150 | head -24
exit $rc
