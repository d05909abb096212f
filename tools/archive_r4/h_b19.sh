#!/bin/bash
# round 4, GPU call 8: 2^19 and 2^18 gates under every layout: window / bitpos 2^15 / 2^17 / 2^19 buckets
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4h
rm -rf $O; mkdir -p $O
cd /tmp
for LG in 19 18; do
for V in "PLONK_MSM_TABLE=window" "PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=15" "PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=17" "PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19"; do
  env $V python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^$LG $V', d['value'], d['kernel_ms_per_prove'], d['proof_blake2b'][:12])"
done
done
