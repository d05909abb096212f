#!/bin/bash
# round 3, GPU call 23: 2^16 (and 2^18) proofs with length-ordered lanes / longer slices / bit-position tables (small-size MSM tail)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3w
rm -rf $O; mkdir -p $O
run() {  # tag, log-gates, env...
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
for LG in 16 18; do
run d$LG $LG X=1
for K in 8 16 32; do
  run o1k${K}_$LG $LG PLONK_MSM_ORDER=1 PLONK_MSM_KSL=$K
  run bpo1k${K}_$LG $LG PLONK_MSM_TABLE=bitpos PLONK_MSM_ORDER=1 PLONK_MSM_KSL=$K
done
run bp_$LG $LG PLONK_MSM_TABLE=bitpos
run bpo0k8_$LG $LG PLONK_MSM_TABLE=bitpos PLONK_MSM_ORDER=0 PLONK_MSM_KSL=8
run o0k8_$LG $LG PLONK_MSM_ORDER=0 PLONK_MSM_KSL=8
run o0k16_$LG $LG PLONK_MSM_ORDER=0 PLONK_MSM_KSL=16
done
