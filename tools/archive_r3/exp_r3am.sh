#!/bin/bash
# round 3, GPU call 39: the driver's bench command on the final build
set -u
O=gpurun_out/r3am
rm -rf $O; mkdir -p $O
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
j = json.loads(open('$O/bench_default_line.json').read().strip().splitlines()[-1])
for k in ('value', 'kernel_ms_per_prove', 'msm_mscalar_per_s', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_2p16', 'prove_ms_host_wires_pinned', 'extras_error'):
    print(k, j.get(k))
print(j['roofline'])
print('cpu', j.get('cpu_baseline', {}).get('value'), j.get('cpu_baseline', {}).get('proof_matches_gpu'))
PY
