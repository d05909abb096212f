#!/bin/bash
# round 3, GPU call 2: bit-position tables (256 rows, width-17 NAF) vs window tables: parity of the MSM / prover tests in both
# modes, then prove() at 2^20 / 2^16 with either table, same box; fixed rates32.  Output: gpurun_out/r3b/
set -u
O=gpurun_out/r3b
rm -rf $O; mkdir -p $O
timeout 120 build/ubench/rates32 > $O/rates32.txt 2>&1
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_msm_variants.py"
timeout 600 python -m pytest $T -m gpu -x -q > $O/tests_bitpos.log 2>&1; echo "bitpos tests rc=$?" | tee -a $O/tests_bitpos.log
tail -4 $O/tests_bitpos.log
PLONK_MSM_TABLE=window timeout 600 python -m pytest $T -m gpu -x -q > $O/tests_window.log 2>&1; echo "window tests rc=$?" | tee -a $O/tests_window.log
tail -4 $O/tests_window.log
B="python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2"
run() { # name, extra args, env...
  local name=$1; shift
  local extra=$1; shift
  env "$@" timeout 200 $B $extra > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'], 'setup', j['config']['setup_s'])
except Exception as e:
    print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run window20 "" PLONK_MSM_TABLE=window
run bitpos20 "" PLONK_MSM_TABLE=bitpos
run bitpos20_ksl64 "" PLONK_MSM_TABLE=bitpos PLONK_MSM_KSL=64
run window20b "" PLONK_MSM_TABLE=window
run bitpos20b "" X=1
run window16 "--log-gates 16 --steps 20" PLONK_MSM_TABLE=window
run bitpos16 "--log-gates 16 --steps 20" PLONK_MSM_TABLE=bitpos
run bl_window "--profile bench-like" PLONK_MSM_TABLE=window
run bl_bitpos "--profile bench-like" PLONK_MSM_TABLE=bitpos
grep "waves/SIMD=2" $O/rates32.txt
