#!/bin/bash
# GPU box: A/B of library builds (build/variants/*.so, see tools notes in DESIGN §4.2): prove ms + MSM kernel split
# at 2^20 gates and prove ms at 2^16 gates
for so in build/variants/libplonk_*.so; do
  export PLONK_HIP_LIB=$PWD/$so
  for lg in ${VARIANT_SIZES:-20 16}; do
    python bench.py --log-gates $lg --steps $((lg == 20 ? 10 : 30)) --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print('$so 2^$lg prove', d['value'], 'acc', k['msm_accumulate'], 'other', k['msm_other'], d['proof_blake2b'][:8])"
  done
done
