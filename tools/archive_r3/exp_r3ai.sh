#!/bin/bash
# round 3, GPU call 35: quotient kernel with all loads issued first (2 waves/SIMD) vs loads at use (4 waves/SIMD)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3ai
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_prover.py tests/test_gpu_prove_sizes.py -x -q -m "gpu and not slow" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() {
  local tag=$1 lg=$2; shift 2
  local extra=""
  if [ "${PROFILE:-}" != "" ]; then extra="--profile $PROFILE"; fi
  env "$@" timeout 300 python bench.py --log-gates $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline $extra > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
    print('$tag', j['value'], j.get('kernel_ms_per_prove'), j.get('roofline_quotient', {}).get('frac'), j.get('proof_blake2b'))
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
run inl20 20 PLONK_QUOTIENT_LOADS=inline
run bat20 20 X=1
run inl20b 20 PLONK_QUOTIENT_LOADS=inline
run bat20b 20 X=1
run inl16 16 PLONK_QUOTIENT_LOADS=inline
run bat16 16 X=1
PROFILE=widgets run inl20w 20 PLONK_QUOTIENT_LOADS=inline
PROFILE=widgets run bat20w 20 X=1
run inl22 22 PLONK_QUOTIENT_LOADS=inline
run bat22 22 X=1
