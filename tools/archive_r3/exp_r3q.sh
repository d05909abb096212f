#!/bin/bash
# round 3, GPU call 17: list-driven bucket sums on a small grid; bit-position tables at 2^22 (table budget: half of the free HBM)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3q
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py"
PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19 timeout 600 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests_nb19.log 2>&1; echo "nb19 small tests rc=$?"; tail -3 $O/tests_nb19.log
B="python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2"
run() { # name, extra args, env...
  local name=$1; shift
  local extra=$1; shift
  env "$@" timeout 300 $B $extra > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'][:8], 'rows', j['roofline'].get('table_rows'), 'setup', j['config']['setup_s'])
except Exception as e:
    print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run nb15 "" PLONK_MSM_BUCKETS=15
run nb19 "" X=1
run nb15b "" PLONK_MSM_BUCKETS=15
run nb19b "" X=1
run p22_nb19 "--log-gates 22 --steps 3 --warmup 1" X=1
run p22_window "--log-gates 22 --steps 3 --warmup 1" PLONK_MSM_TABLE=window
run p21_nb19 "--log-gates 21 --steps 4 --warmup 1" X=1
run p21_nb15 "--log-gates 21 --steps 4 --warmup 1" PLONK_MSM_BUCKETS=15
run p18_default "--log-gates 18 --steps 10 --warmup 2" X=1
run p18_window "--log-gates 18 --steps 10 --warmup 2" PLONK_MSM_TABLE=window
run p19_default "--log-gates 19 --steps 10 --warmup 2" X=1
run p19_nb19 "--log-gates 19 --steps 10 --warmup 2" PLONK_MSM_BUCKETS=19
run p19_window "--log-gates 19 --steps 10 --warmup 2" PLONK_MSM_TABLE=window
