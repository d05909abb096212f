#!/bin/bash
# GPU box, round 6 second session: slice length x lane order at the small sizes (2^16 proof, the 2^17 / 2^18 slices of a
# W = 8 / W = 4 rank), after round 4's dense_quad bucket sums changed what a second slice costs.  One JSON line per run.
out=${1:-gpurun_out/r6b/small_sweep.jsonl}
mkdir -p "$(dirname "$out")"
: > "$out"
for L in 16 17 18; do
  for cfg in "0 -" "4 -" "8 -" "16 -" "32 -" "16 1" "32 1" "32 0"; do
    set -- $cfg
    K=$1; O=$2
    unset PLONK_MSM_KSL PLONK_MSM_ORDER
    [ "$K" != 0 ] && export PLONK_MSM_KSL=$K
    [ "$O" != - ] && export PLONK_MSM_ORDER=$O
    python bench.py --log-gates $L --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print(json.dumps({'log_gates': $L, 'ksl': $K, 'order': '$O', 'prove_ms': d['value'], 'accumulate': k['msm_accumulate'], 'other': k['msm_other'], 'digest': d.get('proof_blake2b')}))" >> "$out"
  done
done
cat "$out"
