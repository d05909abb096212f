#!/bin/bash
# GPU box: lanes in order of slice length at 16-entry slices (2^17-point MSMs: a 2^17-gate proof, the slices of a rank of 8 at
# 2^20 gates) — PLONK_MSM_ORDER=1 against the rule (ordered only from 32-entry slices on), two repetitions, same box
out=${1:-gpurun_out/r6b/order17}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in 1 2 3; do
  for o in - 1; do
    if [ $o = - ]; then unset PLONK_MSM_ORDER; else export PLONK_MSM_ORDER=$o; fi
    for lg in 16 17 18; do
    python bench.py --log-gates $lg --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print(json.dumps({'log_gates': $lg, 'order': '$o', 'prove_ms': d['value'], 'accumulate': k['msm_accumulate'], 'other': k['msm_other']}))"
    done
    python tools/rank_alone.py 20 10 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'order': '$o', 'rank_alone_W8_2p20': d['prove_ms_rank_alone'], 'kernel_ms': d['kernel_ms']}))"
    python tools/rank_alone.py 20 10 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'order': '$o', 'rank_alone_W4_2p20': d['prove_ms_rank_alone'], 'kernel_ms': d['kernel_ms']}))"
  done
done | tee $out/order17.jsonl
