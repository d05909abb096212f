#!/bin/bash
# round 3, GPU call 36: lanes per sum in msm_rowcol_tp (2^19-bucket tail): new rule 4/8/16 vs old 8/16/32 (PLONK_MSM_LPS=1) vs fixed
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3aj
rm -rf $O; mkdir -p $O
PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19 timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -m gpu -k "basic or edge or skew or small_scalars or doubling" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
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
run l4_20 20 PLONK_MSM_LPS=4
run l8_20 20 PLONK_MSM_LPS=8
run l16_20 20 PLONK_MSM_LPS=16
run old22 22 PLONK_MSM_LPS=1
run new22 22 X=1
