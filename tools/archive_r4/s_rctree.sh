#!/bin/bash
# round 4, GPU call: row / column partial sums combined by quads (msm_rowcol_tp QT): parity, A/B against the lane tree
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_msm_variants.py -x -q -m gpu -k "19" 2>&1 | tail -2
cd /tmp
for LG in 20 21; do
for V in lane quad lane quad; do
  PLONK_MSM_RCTREE=$V python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^$LG $V', d['value'], d['kernel_ms_per_prove']['msm_other'], d['proof_blake2b'][:12])"
  [ $LG -eq 21 ] && [ $V = quad ] && break
done
done
