#!/bin/bash
# round 3, GPU call 16: large-bucket variant, fourth cut (aggregated list appends, 128-slice heavy segments on 512 workers,
# slice length by expected bucket size) with 2^19 (main library) and 2^18 buckets (build/variants/libplonk_nb18.so)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_prove_sizes.py"
PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19 timeout 600 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests_nb19.log 2>&1; echo "nb19 small tests rc=$?"; tail -4 $O/tests_nb19.log
PLONK_HIP_LIB=$PWD/build/variants/libplonk_nb18.so PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=18 timeout 600 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests_nb18.log 2>&1; echo "nb18 small tests rc=$?"; tail -4 $O/tests_nb18.log
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
run nb18 "" PLONK_HIP_LIB=$PWD/build/variants/libplonk_nb18.so
run nb15b "" PLONK_MSM_BUCKETS=15
run nb19b "" X=1
run nb18b "" PLONK_HIP_LIB=$PWD/build/variants/libplonk_nb18.so
run bl_nb19 "--profile bench-like" X=1
run bl_nb18 "--profile bench-like" PLONK_HIP_LIB=$PWD/build/variants/libplonk_nb18.so
run wd_nb19 "--profile widgets" X=1
run wd_nb18 "--profile widgets" PLONK_HIP_LIB=$PWD/build/variants/libplonk_nb18.so
run p22_nb15 "--log-gates 22 --steps 3 --warmup 1" PLONK_MSM_BUCKETS=15
run p22_nb19 "--log-gates 22 --steps 3 --warmup 1" X=1
cd /tmp && export TMPDIR=/tmp
for lib in nb19 nb18; do
  L=$GRAFT_REPO_ROOT/plonk_amd/lib/libplonk_hip.so; [ $lib = nb18 ] && L=$GRAFT_REPO_ROOT/build/variants/libplonk_nb18.so
  PLONK_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lib -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 > $O/prof_$lib.log 2>&1
  f=$(find $O/prof_$lib -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$lib.csv
  find $O/prof_$lib -name "*kernel_trace.csv" -delete; find $O/prof_$lib -name "*.db" -delete
  echo "== $lib"
  python - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats_$lib.csv')))[:45]:
    n=r['Name'].split('(')[0].replace('void ','').replace('plonk::','')
    if 'msm' in n: print('%-55s %5s %9.3f ms %9.1f us' % (n[:55], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
done
