#!/bin/bash
# GPU box: slice-length sweep of the MSM for small per-rank sizes (tuning of msm_ksl, msm.hip)
for L in 16 17 18; do
  for K in 0 4 8 16 32; do
    if [ $K = 0 ]; then unset PLONK_MSM_KSL; else export PLONK_MSM_KSL=$K; fi
    python bench.py --log-gates $L --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print('L=$L ksl=$K prove', d['value'], 'acc', k['msm_accumulate'], 'other', k['msm_other'])"
  done
done
