#!/bin/bash
# round 3, GPU call 5: fused heavy-segment workers + LDS-parked recoding: parity (both table modes), then the same-box A/B of
# the table modes on the three workloads; PCIe duplex microbenchmark.  Output: gpurun_out/r3e/
set -u
O=gpurun_out/r3e
rm -rf $O; mkdir -p $O
timeout 60 build/ubench/pcie > $O/pcie.txt 2>&1; cat $O/pcie.txt
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_msm_variants.py"
timeout 600 python -m pytest $T tests/test_gpu_prove_sizes.py -m "gpu and not slow" -x -q > $O/tests_bitpos.log 2>&1; echo "bitpos tests rc=$?"; tail -3 $O/tests_bitpos.log
PLONK_MSM_TABLE=window timeout 600 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests_window.log 2>&1; echo "window tests rc=$?"; tail -3 $O/tests_window.log
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
run window20 "" PLONK_MSM_TABLE=window
run bitpos20 "" PLONK_MSM_TABLE=bitpos
run window20b "" PLONK_MSM_TABLE=window
run bitpos20b "" PLONK_MSM_TABLE=bitpos
run bl_window "--profile bench-like" PLONK_MSM_TABLE=window
run bl_bitpos "--profile bench-like" PLONK_MSM_TABLE=bitpos
run wd_window "--profile widgets" PLONK_MSM_TABLE=window
run wd_bitpos "--profile widgets" PLONK_MSM_TABLE=bitpos
run window16 "--log-gates 16 --steps 20" PLONK_MSM_TABLE=window
run bitpos16 "--log-gates 16 --steps 20" PLONK_MSM_TABLE=bitpos
