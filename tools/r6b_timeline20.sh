#!/bin/bash
# round 6 second session: kernel timeline (with gaps) of the last proof at 2^20 gates
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b/timelines
mkdir -p $O
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for lg in ${1:-20}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$lg -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --log-gates $lg --steps 5 --warmup 2 > $O/bench_$lg.log 2>&1
  python $R/tools/timeline.py $(find $O/t$lg -name "bench_kernel_trace.csv" | head -1) > $O/timeline_2p$lg.txt 2>&1
  cp $(find $O/t$lg -name "bench_kernel_stats.csv" | head -1) $O/kernel_stats_2p$lg.csv
  tail -1 $O/timeline_2p$lg.txt; grep -h metric $O/bench_$lg.log | cut -c1-120
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
