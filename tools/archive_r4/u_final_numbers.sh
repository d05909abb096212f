#!/bin/bash
# round 4, last GPU call: smoke(), rank-alone table and phase table on the final build
set -u
R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp
python $R/tools/rank_alone.py 20 6 1,2,4,8 2>/dev/null | tee $R/gpurun_out/rank_alone_final_2p20.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('rank_alone 2^20 W=%d' % d['world'], d['prove_ms_rank_alone'], d['kernel_ms'])"
python $R/tools/rank_alone.py 22 3 8 2>/dev/null | tee $R/gpurun_out/rank_alone_final_2p22.jsonl | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^22 W=8', d['prove_ms_rank_alone'], d['kernel_ms'])"
python $R/tools/rank_alone.py 16 20 1,8 2>/dev/null | tee $R/gpurun_out/rank_alone_final_2p16.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('rank_alone 2^16 W=%d' % d['world'], d['prove_ms_rank_alone'])"
python $R/tools/msm_phases.py 12 16 20 2>/dev/null | tee $R/gpurun_out/phases_final.jsonl | cut -c1-600
