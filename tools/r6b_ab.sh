#!/bin/bash
# GPU box, round 6 second session — same-box A/B of library builds: build/variants/libplonk_*.so (the baseline built from an
# earlier commit in a worktree) against the tree's own library, interleaved, two repetitions.
#   part 1: prove() at the small sizes + 2^20 with the host-time slots (tools/host_gaps.py)
#   part 2: one rank of 8 alone at 2^20 gates (tools/rank_alone.py)
out=${1:-gpurun_out/r6b/ab}
sizes=${AB_SIZES:-12 16 17 18 20}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
libs="default $(ls build/variants/libplonk_*.so 2>/dev/null)"
for rep in 1 2; do
  for lib in $libs; do
    if [ $lib = default ]; then unset PLONK_HIP_LIB; else export PLONK_HIP_LIB=$PWD/$lib; fi
    python tools/host_gaps.py $sizes 2>>$out/err.txt
  done
done | tee $out/prove_ab.jsonl
if [ -z "$AB_NO_RANK" ]; then
for rep in 1 2; do
  for lib in $libs; do
    if [ $lib = default ]; then unset PLONK_HIP_LIB; else export PLONK_HIP_LIB=$PWD/$lib; fi
    python tools/rank_alone.py 20 10 8 2>>$out/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': '$lib', 'rank_alone_W8_2p20': d['prove_ms_rank_alone'], 'kernel_ms': d['kernel_ms']}))"
  done
done | tee $out/rank8_ab.jsonl
fi
