#!/bin/bash
# round 4, GPU call 14: the benchmarked build under rocprofv3 (kernel stats + PMC passes), timelines at 2^20 / 2^16, the driver's bench command
set -u
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_bench.sh r04n --no-extras > gpurun_out/prof_r04n.log 2>&1; tail -3 gpurun_out/prof_r04n.log | cut -c1-300
python tools/timeline.py gpurun_out/prof_r04n/trace/bench_kernel_trace.csv > gpurun_out/prof_r04n/timeline_2p20.txt 2>&1; head -1 gpurun_out/prof_r04n/timeline_2p20.txt; tail -1 gpurun_out/prof_r04n/timeline_2p20.txt
find gpurun_out/prof_r04n -name "*.db" -delete
find gpurun_out/prof_r04n -name "*kernel_trace.csv" -size +20M -delete
cd /tmp
( time python $R/bench.py > $R/gpurun_out/prof_r04n/bench_default_line.json 2> $R/gpurun_out/prof_r04n/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open('$R/gpurun_out/prof_r04n/bench_default_line.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'prove_ms_2p12', 'prove_ms_2p16', 'prove_ms_2p22', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_host_wires_pinned')})
print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['hbm_frac'], d['kernel_ms_per_prove'])
PY
