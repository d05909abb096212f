#!/bin/bash
# GPU box: A/B of library builds (build/variants/*.so, see tools notes in DESIGN §4.2): prove ms + MSM kernel split
for so in build/variants/libplonk_*.so; do
  export PLONK_HIP_LIB=$PWD/$so
  for rep in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print('$so prove', d['value'], 'acc', k['msm_accumulate'], 'other', k['msm_other'], d['proof_blake2b'][:8])"
  done
done
