#!/bin/bash
# NOTE: PLONK_MSM_TIDY was an experimental switch of round 6, removed after this measurement (< 1 %: profiles/r06/tidy_ab.jsonl); the script is kept as the record of how the A/B was run.
# VERDICT r5 item 2 (bounded attempt at the per-group constant of small proofs and ranks), same-box A/B:
#   PLONK_MSM_TIDY=0  the bucket sort's counters cleared by three hipMemsetAsync launches per group (rounds 2-5)
#   default           cleared inside msm_coarse_scan_kernel (three launches fewer in a chain of ~10 per commitment group)
out=${1:-gpurun_out/r06e}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6
for rep in 1 2; do
for lg in 12 16 18; do
for v in 0 1; do
  PLONK_MSM_TIDY=$v python bench.py --log-gates $lg --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>>$out/ab_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'log_gates': $lg, 'PLONK_MSM_TIDY': $v, 'prove_ms': d['value'], 'kernel_ms': d['kernel_ms_per_prove'], 'proof': d['proof_blake2b']}))"
done; done; done | tee $out/tidy_ab.jsonl
for v in 0 1 0 1; do
  PLONK_MSM_TIDY=$v python tools/rank_alone.py 20 10 8 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'rank_alone_W8_2p20': d['prove_ms_rank_alone'], 'PLONK_MSM_TIDY': $v, 'kernel_ms': d['kernel_ms']}))"
done | tee $out/tidy_ab_rank8.jsonl
