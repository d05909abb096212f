#!/bin/bash
# round 3, GPU call 12: the 2^19-bucket variant (width-21 NAF, one lane per bucket, throughput tail): parity, then A/B
set -u
O=gpurun_out/r3l
rm -rf $O; mkdir -p $O
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py"
PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19 timeout 600 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests_nb19.log 2>&1; echo "nb19 small tests rc=$?"; tail -15 $O/tests_nb19.log
timeout 600 python -m pytest $T tests/test_gpu_msm_variants.py -m "gpu and not slow" -x -q > $O/tests_default.log 2>&1; echo "default tests rc=$?"; tail -5 $O/tests_default.log
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
run bl_nb15 "--profile bench-like" PLONK_MSM_BUCKETS=15
run bl_nb19 "--profile bench-like" X=1
run wd_nb15 "--profile widgets" PLONK_MSM_BUCKETS=15
run wd_nb19 "--profile widgets" X=1
