#!/bin/bash
# round 4, GPU call 11: host worker pool for the commitment chains: parity + same-box A/B at small sizes
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4k
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_prover.py tests/test_c_example.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp
for LG in 12 16 14 20; do
  for V in 0 3 0 3; do
    PLONK_HOST_THREADS=$V python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps $([ $LG -ge 19 ] && echo 10 || echo 40) --warmup 3 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^$LG threads=$V', d['value'], d['proof_blake2b'][:12])"
    [ $LG -eq 20 ] && [ $V -eq 3 ] && break
  done
done
for V in 0 3; do
PLONK_HOST_THREADS=$V python $R/tools/rank_alone.py 20 8 8 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^20 W=8 threads=$V', d['prove_ms_rank_alone'])"
done
