#!/bin/bash
# round 3, GPU call 22: kernel timelines of one proof at 2^16 and 2^12 on the final build (small-size latency)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3v
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for LG in 16 12 18; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$LG -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps 5 --warmup 2 > $O/bench_$LG.log 2>&1
  T=$(find $O/t$LG -name "*kernel_trace.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/timeline.py $T > $O/timeline_$LG.txt
  head -1 $O/timeline_$LG.txt; tail -1 $O/timeline_$LG.txt
  find $O/t$LG -name "*.db" -delete
done
