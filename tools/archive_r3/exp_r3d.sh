#!/bin/bash
# round 3, GPU call 4: per-kernel rocprofv3 stats of prove() in either table mode (csv), the new bench line entries and the
# full-duplex plonk_ntt_batch (default bench run).  Output: gpurun_out/r3d/
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
j = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
for k in ('value', 'kernel_ms_per_prove', 'roofline', 'roofline_quotient', 'roofline_ntt', 'leaf_ms', 'prove_ms_bench_like', 'prove_ms_all_widgets_pi', 'prove_ms_2p16', 'prove_ms_host_wires_pinned', 'extras_error'):
    print(k, j.get(k))
print('cpu', j.get('cpu_baseline', {}).get('value'), j.get('cpu_baseline', {}).get('proof_matches_gpu'))
PY
cd /tmp && export TMPDIR=/tmp
for mode in window bitpos; do
  PLONK_MSM_TABLE=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 > $O/prof_$mode.log 2>&1
  f=$(find $O/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode $f"; head -30 "$f" | cut -d, -f1-5
  cp "$f" $O/kernel_stats_$mode.csv
  find $O/prof_$mode -name "*kernel_trace.csv" -delete; find $O/prof_$mode -name "*.db" -delete
done
