#!/bin/bash
# kernel trace + last-proof timeline at 2^16 and 2^20 (no PMC); output under gpurun_out/exp_trace/
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/exp_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in 16 20; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$L -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --log-gates $L --steps 5 --warmup 2 > $O/bench_$L.log 2>&1
  T=$(find $O/t$L -name "bench_kernel_trace.csv" | head -1)
  python $R/tools/timeline.py $T > $O/timeline_$L.txt 2>&1
  cp $(find $O/t$L -name "bench_kernel_stats.csv" | head -1) $O/stats_$L.csv
  rm -rf $O/t$L
  tail -1 $O/bench_$L.log | cut -c1-200
done
