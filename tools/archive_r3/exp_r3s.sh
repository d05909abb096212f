#!/bin/bash
# round 3, GPU call 19: NAF recoding with shared top digits: parity, A/B against 2^15 buckets, kernel stats
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3t
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_prove_sizes.py tests/test_gpu_msm_variants.py"
timeout 900 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
B="python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2"
run() { # name, extra args, env...
  local name=$1; shift
  local extra=$1; shift
  env "$@" timeout 300 $B $extra > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'][:8])
except Exception as e:
    print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run window "" PLONK_MSM_TABLE=window

run nb19 "" X=1

run nb19b "" X=1
run bl_nb19 "--profile bench-like" X=1
run wd_nb19 "--profile widgets" X=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
python - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats.csv')))[:45]:
    n=r['Name'].split('(')[0].replace('void ','').replace('plonk::','')
    if 'msm' in n: print('%-55s %5s %9.3f ms %9.1f us' % (n[:55], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
