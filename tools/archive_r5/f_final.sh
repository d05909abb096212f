#!/bin/bash
# round 5, GPU call f: whole GPU suite (wall time after the trims), then the benchmarked build under rocprofv3 (stats + PMC),
# timeline, one A/B of the accumulate workgroup size, and the driver's default bench line
set -u
R=$GRAFT_REPO_ROOT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/prof_r05x
python __graft_entry__.py > gpurun_out/prof_r05x/build.log 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu --durations=30 ) > gpurun_out/prof_r05x/gpu_suite.txt 2>&1
tail -45 gpurun_out/prof_r05x/gpu_suite.txt
python tools/archive_r5/quick.py acc_wg_128 --steps 10 --warmup 2 > gpurun_out/prof_r05x/ab_acc_wg.jsonl
PLONK_MSM_ACC_WG=64 python tools/archive_r5/quick.py acc_wg_64 --steps 10 --warmup 2 >> gpurun_out/prof_r05x/ab_acc_wg.jsonl
python tools/archive_r5/quick.py acc_wg_128_again --steps 10 --warmup 2 >> gpurun_out/prof_r05x/ab_acc_wg.jsonl
cat gpurun_out/prof_r05x/ab_acc_wg.jsonl
bash tools/profile_bench.sh r05x --no-extras > gpurun_out/prof_r05x.log 2>&1
python tools/timeline.py gpurun_out/prof_r05x/trace/bench_kernel_trace.csv > gpurun_out/prof_r05x/timeline_2p20.txt 2>&1; head -1 gpurun_out/prof_r05x/timeline_2p20.txt
find gpurun_out/prof_r05x -name "*.db" -delete
cd /tmp
( time python $R/bench.py > $R/gpurun_out/prof_r05x/bench_default_line.json 2> $R/gpurun_out/prof_r05x/bench_default.err ) 2>&1 | grep real
tail -3 $R/gpurun_out/prof_r05x/bench_default.err
python - <<PY
import json
d = json.loads(open('$R/gpurun_out/prof_r05x/bench_default_line.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'prove_ms_2p12', 'prove_ms_2p16', 'prove_ms_2p22', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_host_wires_pinned')})
print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['hbm_frac'], d['roofline']['traffic_source'], d['kernel_ms_per_prove'])
print([ (r['m'], r['batch'], r['scalars'], r['device_ms'], r['mscalar_per_s'], r['kernel']) for r in d.get('msm_micro', [])])
print([ (r['log_size'], r['transform'], r['ms']) for r in d.get('ntt_micro', [])])
print({k: (d[k].get('value'), d[k].get('proof_matches_gpu')) for k in d if k.startswith('cpu_baseline')})
print({k: d[k] for k in d if k.endswith('_error')})
PY
