#!/bin/bash
# round 4, GPU call 15: differential soak of the new layouts against the C oracle + the V2 transcript test
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_prover.py -x -q -m gpu -k "legacy" 2>&1 | tail -2
for V in "PLONK_MSM_TABLE=halfpos" "PLONK_MSM_TABLE=halfpos PLONK_MSM_BUCKETS=19" "PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19" "PLONK_MSM_BSUM=lane"; do
  env $V timeout 900 python tools/soak_parity.py 24 $RANDOM 2>&1 | tail -1 | cut -c1-200 | sed "s/^/[$V] /"
done
