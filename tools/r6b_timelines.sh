#!/bin/bash
# round 6 second session: kernel timelines (with gaps) of the last proof at 2^12 / 2^16 and of a rank of 8 at 2^20
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b/timelines
mkdir -p $O
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for lg in 12 16; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t$lg -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --log-gates $lg --steps 5 --warmup 2 > $O/bench_$lg.log 2>&1
  python $R/tools/timeline.py $(find $O/t$lg -name "bench_kernel_trace.csv" | head -1) > $O/timeline_2p$lg.txt 2>&1
  tail -1 $O/timeline_2p$lg.txt; grep -h metric $O/bench_$lg.log | cut -c1-120
done
rocprofv3 --kernel-trace --output-format csv -d $O/trank -o bench -- python $R/tools/rank_alone.py 20 3 8 > $O/rank8.log 2>&1
python $R/tools/timeline.py $(find $O/trank -name "bench_kernel_trace.csv" | head -1) > $O/timeline_rank8_2p20.txt 2>&1
tail -1 $O/timeline_rank8_2p20.txt; cat $O/rank8.log | tail -2
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
