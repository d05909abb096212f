#!/bin/bash
# round 4, GPU call 3: un-profiled per-phase kernel times of the MSM pipeline (hipEvents) at small sizes, lane vs quad bucket sums
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
rm -rf $O; mkdir -p $O
cd /tmp
for V in lane quad; do
  PLONK_MSM_BSUM=$V python $R/tools/msm_phases.py 16 12 17 > $O/phases_$V.jsonl 2> $O/phases_$V.err
  echo "== $V"; cat $O/phases_$V.jsonl
done
python $R/tools/msm_phases.py 20 > $O/phases_20.jsonl 2>> $O/phases_quad.err; cat $O/phases_20.jsonl
