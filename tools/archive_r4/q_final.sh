#!/bin/bash
# round 4, final GPU call: the whole GPU suite as the driver runs it, then the benchmarked build under rocprofv3 (stats + PMC), timeline, default bench line
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r4q
( time python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r4q/pytest.log 2>&1 ) 2>&1 | grep real
echo "pytest rc=$?"; tail -14 gpurun_out/r4q/pytest.log | cut -c1-160
bash tools/profile_bench.sh r04q --no-extras > gpurun_out/prof_r04q.log 2>&1
python tools/timeline.py gpurun_out/prof_r04q/trace/bench_kernel_trace.csv > gpurun_out/prof_r04q/timeline_2p20.txt 2>&1; head -1 gpurun_out/prof_r04q/timeline_2p20.txt; tail -1 gpurun_out/prof_r04q/timeline_2p20.txt
find gpurun_out/prof_r04q -name "*.db" -delete
cd /tmp
( time python $R/bench.py > $R/gpurun_out/prof_r04q/bench_default_line.json 2> $R/gpurun_out/prof_r04q/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open('$R/gpurun_out/prof_r04q/bench_default_line.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'prove_ms_2p12', 'prove_ms_2p16', 'prove_ms_2p22', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_host_wires_pinned')})
print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['hbm_frac'], d['roofline']['traffic_source'], d['kernel_ms_per_prove'])
PY
