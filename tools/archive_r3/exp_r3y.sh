#!/bin/bash
# round 3, GPU call 25: standalone transform times + per-kernel stats with / without direct inter-pass twiddle tables
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3y
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 0 1; do
  PLONK_NTT_DIRECT=$M timeout 400 python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ex$M.json 2> $O/ex$M.err
  python - <<PY
import json
j = json.loads(open('$O/ex$M.json').read().strip().splitlines()[-1])
print('direct=$M', j['value'], {k: v['ms'] for k, v in j.get('roofline_ntt', {}).get('transforms', {}).items()}, j.get('leaf_ms'))
PY
  PLONK_NTT_DIRECT=$M rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$M -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_t$M.log 2>&1
  grep -h "ntt_pass" $O/t$M/*kernel_stats.csv | cut -d, -f1-5
  find $O/t$M -name "*.db" -delete; find $O/t$M -name "*kernel_trace.csv" -delete
done
