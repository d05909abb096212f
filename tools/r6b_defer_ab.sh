#!/bin/bash
# GPU box: side-stream transforms started with their commitment group (default from 2^19 gates on) or after its accumulation
# (PLONK_SIDE_DEFER=1: rounds 1 and 2, =2: round 1 only), 2^19 / 2^20 gates, same box, two repetitions
out=${1:-gpurun_out/r6b/defer}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in 1 2; do
  for dfr in - 1 2; do
    for lg in 19 20; do
      if [ $dfr = - ]; then unset PLONK_SIDE_DEFER; else export PLONK_SIDE_DEFER=$dfr; fi
      python bench.py --log-gates $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print(json.dumps({'side_defer': '$dfr', 'log_gates': $lg, 'prove_ms': d['value'], 'accumulate': k['msm_accumulate'], 'other': k['msm_other']}))"
    done
  done
done | tee $out/defer_ab.jsonl
