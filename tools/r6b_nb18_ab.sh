#!/bin/bash
# GPU box: the "large" MSM variant with 2^18 instead of 2^19 buckets (build/variants/libplonk_nb18.so, tools/build_variants.sh 18)
# where a commitment has 2^19 points — a 2^19-gate proof, the range-sharded groups of a rank of 2 at 2^20 — and at 2^20 for
# reference; same box, two repetitions
out=${1:-gpurun_out/r6b/nb18}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in 1 2; do
  for lib in default build/variants/libplonk_nb18.so; do
    if [ $lib = default ]; then unset PLONK_HIP_LIB; else export PLONK_HIP_LIB=$PWD/$lib; fi
    for lg in 19 20; do
      python bench.py --log-gates $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms_per_prove']
print(json.dumps({'lib': '$lib', 'log_gates': $lg, 'prove_ms': d['value'], 'accumulate': k['msm_accumulate'], 'other': k['msm_other'], 'digest': d.get('proof_blake2b')}))"
    done
    python tools/rank_alone.py 20 10 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': '$lib', 'rank_alone_W2_2p20': d['prove_ms_rank_alone'], 'kernel_ms': d['kernel_ms']}))"
  done
done | tee $out/nb18_ab.jsonl
