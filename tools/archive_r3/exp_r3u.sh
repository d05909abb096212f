#!/bin/bash
# round 3, GPU call 21: final build — whole GPU suite, default bench line, rocprofv3 stats + PMC passes, 2^22 / 2^16 / 2^12 lines
set -u
O=gpurun_out/r3u
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -16 $O/gpu_suite.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
j = json.loads(open('$O/bench_default_line.json').read().strip().splitlines()[-1])
for k in ('value', 'kernel_ms_per_prove', 'msm_mscalar_per_s', 'roofline', 'roofline_quotient', 'leaf_ms', 'leaf_error', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_2p16', 'prove_ms_host_wires_pinned', 'extras_error'):
    print(k, j.get(k))
print('ntt', {k: (v['ms'], v['melem_per_s']) for k, v in j.get('roofline_ntt', {}).get('transforms', {}).items()})
print('compile', j.get('compile'))
print('cpu', j.get('cpu_baseline', {}).get('value'), j.get('cpu_baseline', {}).get('proof_matches_gpu'))
PY
timeout 600 python bench.py --log-gates 22 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_2p22_line.json 2> $O/bench_2p22.err; echo "2^22 rc=$?"; cut -c1-200 $O/bench_2p22_line.json
timeout 300 python bench.py --log-gates 16 --steps 30 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_2p16_line.json 2> $O/bench_2p16.err; cut -c1-200 $O/bench_2p16_line.json
timeout 300 python bench.py --log-gates 12 --steps 30 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_2p12_line.json 2> $O/bench_2p12.err; cut -c1-200 $O/bench_2p12_line.json
bash tools/profile_bench.sh r03d --steps 5 --warmup 2 --no-extras > $O/profile.log 2>&1; tail -3 $O/profile.log
find gpurun_out/prof_r03d -name "*.db" -delete
find gpurun_out/prof_r03d -name "*kernel_trace.csv" -size +30M -delete
du -sh gpurun_out/prof_r03d
