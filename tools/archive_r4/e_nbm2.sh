#!/bin/bash
# round 4, GPU call 5: nbm with pair lanes + LPS rule: parity, phases at 2^19 (pair on/off, LPS), kernel trace for the sort
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msm_variants.py -x -q -m gpu -k "17" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp
ph() { python -c "
import json,sys
for l in open('$1'):
    d=json.loads(l); print('$2', d['prove_ms'], 'g4', d['groups_of_4'], 'g12', d['groups_of_1_2'])
"; }
for V in "PLONK_MSM_PAIR=0" "PLONK_MSM_PAIR=1" "X=1" "PLONK_MSM_LPS=8" "PLONK_MSM_LPS=16" "PLONK_MSM_LPS=32" "PLONK_MSM_TABLE=window"; do
  env $V python $R/tools/msm_phases.py 19 > $O/ph_$V.jsonl 2> $O/ph_$V.err; ph $O/ph_$V.jsonl "$V"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t19 -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --log-gates 19 --steps 2 --warmup 1 > $O/trace.log 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('$O/t19/*kernel_stats.csv')[0])))
for r in rows[:40]:
    n = r['Name'].split('(')[0].replace('void plonk::', '').replace('plonk::', '')
    if 'msm' in n: print('%-60s calls %4s avg_us %9.1f' % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
find $O -name "*.db" -delete
