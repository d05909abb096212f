#!/bin/bash
# round 3, GPU call 14: 2^19-bucket variant with multi-workgroup layout kernels, direct bucket writes, tail by group size
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3o
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_prove_sizes.py"
PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19 timeout 600 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests_nb19.log 2>&1; echo "nb19 small tests rc=$?"; tail -4 $O/tests_nb19.log
timeout 600 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests_default.log 2>&1; echo "default tests rc=$?"; tail -3 $O/tests_default.log
B="python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2"
run() { # name, extra args, env...
  local name=$1; shift
  local extra=$1; shift
  env "$@" timeout 200 $B $extra > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'][:8])
except Exception as e:
    print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run nb15 "" PLONK_MSM_BUCKETS=15
run nb19 "" X=1
run nb15b "" PLONK_MSM_BUCKETS=15
run nb19b "" X=1
run bl_nb19 "--profile bench-like" X=1
run wd_nb19 "--profile widgets" X=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_nb19.csv
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
python - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats_nb19.csv')))[:45]:
    n=r['Name'].split('(')[0].replace('void ','').replace('plonk::','')
    if 'msm' in n: print('%-60s %5s %10.3f ms %10.1f us' % (n[:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
