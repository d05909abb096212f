#!/bin/bash
# round 4, last GPU call: the benchmarked build (4-elements-per-lane NTT passes by default) under rocprofv3 (stats + PMC), timeline,
# the driver's default bench line, and the parity tests the NTT change touches that the earlier calls did not run
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_r04x
bash tools/profile_bench.sh r04x --no-extras > gpurun_out/prof_r04x.log 2>&1
python tools/timeline.py gpurun_out/prof_r04x/trace/bench_kernel_trace.csv > gpurun_out/prof_r04x/timeline_2p20.txt 2>&1; head -1 gpurun_out/prof_r04x/timeline_2p20.txt
find gpurun_out/prof_r04x -name "*.db" -delete
cd /tmp
( time python $R/bench.py > $R/gpurun_out/prof_r04x/bench_default_line.json 2> $R/gpurun_out/prof_r04x/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open('$R/gpurun_out/prof_r04x/bench_default_line.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'prove_ms_2p12', 'prove_ms_2p16', 'prove_ms_2p22', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_host_wires_pinned')})
print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['hbm_frac'], d['roofline']['traffic_source'], d['kernel_ms_per_prove'])
print({k: (v['ms'], v['frac']) for k, v in d['roofline_ntt']['transforms'].items()}, d['roofline_ntt']['kernel'])
print(d.get('compile', {}).get('proof_matches_timed_run'), d.get('compile', {}).get('vk_blake2b'), d['cpu_baseline'].get('proof_matches_gpu'))
PY
cd $R
timeout 600 python -m pytest tests/test_gpu_compile.py tests/test_gpu_multirank.py -x -q -m "gpu and not slow" 2>&1 | tail -3
