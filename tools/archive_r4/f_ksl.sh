#!/bin/bash
# round 4, GPU call 6: nbm at 2^19 with finer slices (KSL 32 / 64 vs 128)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
rm -rf $O; mkdir -p $O
cd /tmp
ph() { python -c "
import json,sys
for l in open('$1'):
    d=json.loads(l); print('$2', d['prove_ms'], 'g4', d['groups_of_4'], 'g12', d['groups_of_1_2'])
"; }
for V in "PLONK_MSM_KSL=32" "PLONK_MSM_KSL=64" "PLONK_MSM_KSL=128" "PLONK_MSM_KSL=32 PLONK_MSM_PAIR=0" "PLONK_MSM_KSL=16 PLONK_MSM_PAIR=0"; do
  env $V python $R/tools/msm_phases.py 19 > "$O/ph_$V.jsonl" 2> "$O/ph_$V.err"; ph "$O/ph_$V.jsonl" "$V"
done
