#!/bin/bash
# NOTE: PLONK_MSM_HEAVY_MIN was an experimental switch of this session, removed after the measurement (+-0.1 ms: profiles/r06b/heavy_ab.jsonl); the script is the record of how the A/B was run.
# GPU box: smallest slice count of a "heavy" bucket in the 2^19-bucket variant (PLONK_MSM_HEAVY_MIN; default 16), same box
out=${1:-gpurun_out/r6b/heavy}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in 1 2; do
  for h in 0 4 8; do
    for prof in dense bench-like widgets; do
      if [ $h = 0 ]; then unset PLONK_MSM_HEAVY_MIN; else export PLONK_MSM_HEAVY_MIN=$h; fi
      python bench.py --log-gates 20 --profile $prof --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print(json.dumps({'heavy_min': $h, 'profile': '$prof', 'prove_ms': d['value'], 'accumulate': k['msm_accumulate'], 'other': k['msm_other'], 'digest': d.get('proof_blake2b')}))"
    done
  done
done | tee $out/heavy_ab.jsonl
