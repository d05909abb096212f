#!/bin/bash
# round 3, GPU call 29: lanes per bucket in msm_bucket_sum chosen by occupancy (PLONK_MSM_BSG=2 = the old rule at these sizes)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3ac
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
for LG in 12 14 16 17 18 19; do
  run old_$LG $LG PLONK_MSM_BSG=2
  run new_$LG $LG X=1
  run g1_$LG $LG PLONK_MSM_BSG=1
done
