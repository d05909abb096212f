#!/bin/bash
# GPU box: waves per row / column sum of msm_rowcol_quad (2^15 buckets) for groups of 1 / 2 / >= 3 commitments
# (PLONK_MSM_RCWV=abc; the rule is 421), small proofs and a rank of 8 alone, same box, two repetitions
out=${1:-gpurun_out/r6b/rcwv}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in 1 2; do
  for v in 0 211 221 111 411 222; do
    if [ $v = 0 ]; then unset PLONK_MSM_RCWV; else export PLONK_MSM_RCWV=$v; fi
    for lg in 12 16 17; do
      python bench.py --log-gates $lg --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print(json.dumps({'rcwv': $v, 'log_gates': $lg, 'prove_ms': d['value'], 'other': k['msm_other']}))"
    done
    python tools/rank_alone.py 20 10 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'rcwv': $v, 'rank_alone_W8_2p20': d['prove_ms_rank_alone'], 'other': d['kernel_ms']['msm_other']}))"
  done
done | tee $out/rcwv.jsonl
