#!/bin/bash
# round 5, GPU call i: smoke() and a short default-shaped bench line after the last bench.py edits
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
python bench.py --steps 5 --warmup 1 --no-2p22 > gpurun_out/r5i/bench.json 2> gpurun_out/r5i/bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r5i/bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['value'], r['frac'], r.get('shader_clock_ghz_under_kernel'), r.get('frac_at_measured_clock'), r['traffic_source'], r['valu_int_fraction'])
print({k: d[k] for k in d if k.endswith('_error')}, len(d.get('msm_micro', [])), len(d.get('ntt_micro', [])), d['cpu_baseline'].get('proof_matches_gpu'), d['cpu_baseline_2p16'].get('proof_matches_gpu'))
PY
tail -3 gpurun_out/r5i/bench.err
