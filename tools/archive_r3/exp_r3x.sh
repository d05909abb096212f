#!/bin/bash
# round 3, GPU call 24: direct inter-pass twiddle tables in the NTT passes (PLONK_NTT_DIRECT=0 = two-level product), same box
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3x
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu > $O/ntt_tests.log 2>&1; echo "ntt tests rc=$?"; tail -3 $O/ntt_tests.log
run() {
  local tag=$1 lg=$2; shift 2
  env "$@" timeout 300 python bench.py --log-gates $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
    print('$tag', j['value'], j.get('kernel_ms_per_prove'), {k: v['ms'] for k, v in j.get('roofline_ntt', {}).get('transforms', {}).items()}, j.get('proof_blake2b'))
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
run two20 20 PLONK_NTT_DIRECT=0
run dir20 20 X=1
run two20b 20 PLONK_NTT_DIRECT=0
run dir20b 20 X=1
run two16 16 PLONK_NTT_DIRECT=0
run dir16 16 X=1
run two22 22 PLONK_NTT_DIRECT=0
run dir22 22 X=1
