#!/bin/bash
# round 4, GPU call 9: build A (nbl from 2^18 terms, strided big-bin kernels, sharded side schedule): skew / variant parity, 2^19-2^20 timing,
# rank-alone numbers, timeline of a W = 8 rank, the default bench line end to end
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4i
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_variants.py tests/test_gpu_prove_sizes.py -x -q -m "gpu" -k "not 2p20" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp
for LG in 20 19 16 12; do
  python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps $([ $LG -ge 19 ] && echo 10 || echo 30) --warmup 3 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^$LG', d['value'], d['kernel_ms_per_prove'], d['proof_blake2b'][:12])"
done
python $R/tools/rank_alone.py 20 5 2,4,8 2> $O/ra.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('rank_alone 2^20 W=%d' % d['world'], d['prove_ms_rank_alone'], d['kernel_ms'], d['table_rows'])"
python $R/tools/rank_alone.py 22 3 8 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^22 W=8', d['prove_ms_rank_alone'], d['kernel_ms'], d['table_rows'])"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr8 -o ra -- python $R/tools/rank_alone.py 20 3 8 > $O/tr8.log 2>&1
python $R/tools/timeline.py $(find $O/tr8 -name "*kernel_trace.csv" | head -1) > $O/timeline_rank8_2p20.txt 2>&1; head -1 $O/timeline_rank8_2p20.txt; tail -1 $O/timeline_rank8_2p20.txt
find $O -name "*.db" -delete
( time python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
keys = ['value', 'prove_ms_2p12', 'prove_ms_2p16', 'prove_ms_2p22', 'proof_blake2b_2p22', 'prove_2p22_setup_and_run_s', 'prove_2p22_error', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'extras_error']
print({k: d.get(k) for k in keys})
print('roofline', {k: d['roofline'].get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'hbm_frac', 'traffic_source')})
print('cpu', d.get('cpu_baseline'))
PY
