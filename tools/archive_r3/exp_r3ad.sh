#!/bin/bash
# round 3, GPU call 30: slice length rule below 2^20 — default vs 2x / 4x longer slices (ordered lanes switch on at 32)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3ad
rm -rf $O; mkdir -p $O
run() {
  local tag=$1 lg=$2; shift 2
  env "$@" timeout 200 python bench.py --log-gates $lg --steps 30 --warmup 3 --no-extras --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
    print('$tag', j['value'], j.get('kernel_ms_per_prove'), j.get('proof_blake2b'))
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
# default ksl: 2^12..2^16 -> 4, 2^17 -> 4, 2^18 -> 8, 2^19 -> 16
run d_12 12 X=1; run k8_12 12 PLONK_MSM_KSL=8; run k16_12 12 PLONK_MSM_KSL=16
run d_14 14 X=1; run k8_14 14 PLONK_MSM_KSL=8; run k16_14 14 PLONK_MSM_KSL=16
run d_16 16 X=1; run k8_16 16 PLONK_MSM_KSL=8; run k16_16 16 PLONK_MSM_KSL=16; run d_16b 16 X=1; run k8_16b 16 PLONK_MSM_KSL=8
run d_17 17 X=1; run k8_17 17 PLONK_MSM_KSL=8; run k16_17 17 PLONK_MSM_KSL=16; run k32_17 17 PLONK_MSM_KSL=32
run d_18 18 X=1; run k16_18 18 PLONK_MSM_KSL=16; run k32_18 18 PLONK_MSM_KSL=32; run d_18b 18 X=1; run k32_18b 18 PLONK_MSM_KSL=32
run d_19 19 X=1; run k32_19 19 PLONK_MSM_KSL=32; run k64_19 19 PLONK_MSM_KSL=64; run d_19b 19 X=1; run k32_19b 19 PLONK_MSM_KSL=32
