#!/bin/bash
# round 3, GPU call 10: the build with the safegcd inversions: parity tests, then the measurements that go to profiles/r03b:
# default bench line, rocprofv3 kernel stats + PMC passes (tools/profile_bench.sh), 2^22 and 2^16 lines.
set -u
O=gpurun_out/r3j
rm -rf $O; mkdir -p $O
T="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_compile.py tests/test_gpu_prove_sizes.py tests/test_gpu_soak.py"
timeout 900 python -m pytest $T -m "gpu and not slow" -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
j = json.loads(open('$O/bench_default_line.json').read().strip().splitlines()[-1])
for k in ('value', 'kernel_ms_per_prove', 'msm_mscalar_per_s', 'roofline', 'roofline_quotient', 'leaf_ms', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_2p16', 'prove_ms_host_wires_pinned', 'extras_error'):
    print(k, j.get(k))
print('ntt', {k: (v['ms'], v['melem_per_s']) for k, v in j.get('roofline_ntt', {}).get('transforms', {}).items()})
print('compile', j.get('compile'))
print('cpu', j.get('cpu_baseline', {}).get('value'), j.get('cpu_baseline', {}).get('proof_matches_gpu'))
PY
timeout 600 python bench.py --log-gates 22 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_2p22_line.json 2> $O/bench_2p22.err; echo "2^22 rc=$?"; cut -c1-400 $O/bench_2p22_line.json
timeout 300 python bench.py --log-gates 16 --steps 30 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_2p16_line.json 2> $O/bench_2p16.err; cut -c1-300 $O/bench_2p16_line.json
timeout 300 python bench.py --log-gates 12 --steps 30 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_2p12_line.json 2> $O/bench_2p12.err; cut -c1-300 $O/bench_2p12_line.json
bash tools/profile_bench.sh r03b --steps 5 --warmup 2 --no-extras > $O/profile.log 2>&1; tail -5 $O/profile.log
find gpurun_out/prof_r03b -name "*.db" -delete
find gpurun_out/prof_r03b -name "*kernel_trace.csv" -size +30M -delete
du -sh gpurun_out/prof_r03b
