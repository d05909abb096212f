#!/bin/bash
# GPU box: z committed from its evaluations over the Lagrange-basis key (default) against the coefficient form
# (PLONK_Z_COMMIT=coeff), same library, same box, three repetitions
out=${1:-gpurun_out/r6b/zcommit}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in 1 2 3; do
  for z in - coeff; do
    if [ $z = - ]; then unset PLONK_Z_COMMIT; else export PLONK_Z_COMMIT=$z; fi
    python tools/host_gaps.py 12 16 17 18 20 2>>$out/err.txt | sed "s/^{/{\"z_commit\": \"$z\", /"
  done
done | tee $out/zcommit_ab.jsonl
