#!/bin/bash
# round 3, GPU call 13: per-kernel stats of the 2^19-bucket variant
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3m
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_nb19.csv
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
python - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats_nb19.csv')))[:40]:
    n=r['Name'].split('(')[0].replace('void ','').replace('plonk::','')
    if 'msm' in n or 'nb19' in n: print('%-60s %5s %10.3f ms %10.1f us' % (n[:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
