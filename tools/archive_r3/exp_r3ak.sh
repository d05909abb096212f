#!/bin/bash
# round 3, GPU call 37: msm_rowcol_tp lanes per sum 8/8/16 (new default) vs 8/16/32 (PLONK_MSM_LPS=1) at 2^20, 2^21, 2^22
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3ak
rm -rf $O; mkdir -p $O
run() {
  local tag=$1 lg=$2; shift 2
  env "$@" timeout 300 python bench.py --log-gates $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
    print('$tag', j['value'], j.get('kernel_ms_per_prove'), j.get('proof_blake2b'))
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
run old20 20 PLONK_MSM_LPS=1
run new20 20 X=1
run old20b 20 PLONK_MSM_LPS=1
run new20b 20 X=1
run old22 22 PLONK_MSM_LPS=1
run new22 22 X=1
run old22b 22 PLONK_MSM_LPS=1
run new22b 22 X=1
run old21 21 PLONK_MSM_LPS=1
run new21 21 X=1
