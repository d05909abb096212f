#!/bin/bash
# round 4, GPU call: fold kernel with partial low-row sums (no 256-point H_r chain): parity + timing
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_msm_variants.py -x -q -m gpu -k "19" 2>&1 | tail -2
cd /tmp
for LG in 20 19; do
for V in 1 2; do
  python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^$LG', d['value'], d['kernel_ms_per_prove']['msm_other'], d['proof_blake2b'][:12])"
done
done
python $R/tools/msm_phases.py 20 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('phases 2^20', d['prove_ms'], 'g4', d['groups_of_4'], 'g12', d['groups_of_1_2'])"
