#!/bin/bash
# round 4, GPU call 2: quad-per-bucket bucket sums (dense_quad) + batched-load slices kernel: parity tests, then same-box A/B
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4b
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_msm_variants.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d['value'], {k: round(v, 3) for k, v in d.get('kernel_ms_per_prove', {}).items()}, d.get('proof_blake2b', '')[:16])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
for LG in 16 12 14 17 18 19; do
  for V in lane quad lane quad; do
    PLONK_MSM_BSUM=$V python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps 30 --warmup 3 > $O/b_${LG}_$V.json 2> $O/b_${LG}_$V.err
    line $O/b_${LG}_$V.json "2^$LG $V"
    [ $LG -ne 16 ] && [ $V = quad ] && break
  done
done
for V in lane quad; do
  PLONK_MSM_BSUM=$V python $R/tools/rank_alone.py 20 5 8 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^20 W=8 $V', d['prove_ms_rank_alone'], d['kernel_ms'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t16 -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --log-gates 16 --steps 2 --warmup 1 > $O/trace.log 2>&1
python $R/tools/timeline.py $(find $O/t16 -name "*kernel_trace.csv" | head -1) > $O/timeline_16.txt; grep -E "bucket_sum|slices|last proof|device busy" $O/timeline_16.txt
find $O -name "*.db" -delete
