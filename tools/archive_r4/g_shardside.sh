#!/bin/bash
# round 4, GPU call 7: sharded prover with the side-stream schedule: multirank parity, rank-alone A/B (PLONK_SHARD_SIDE=0/1)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4g
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q -m "gpu and not slow" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp
for LG in 20 16; do
for W in 8 2; do
  for V in 0 1; do
    PLONK_SHARD_SIDE=$V python $R/tools/rank_alone.py $LG 5 $W 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^$LG W=$W side=$V', d['prove_ms_rank_alone'], d['kernel_ms'], d['table_rows'])"
  done
done
done
PLONK_SHARD_SIDE=0 python $R/tools/rank_alone.py 22 3 8 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^22 W=8 side=0', d['prove_ms_rank_alone'], d['kernel_ms'], d['table_rows'])"
python $R/tools/rank_alone.py 22 3 8 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^22 W=8 side=1', d['prove_ms_rank_alone'], d['kernel_ms'], d['table_rows'])"
python $R/bench.py --no-cpu-baseline --no-extras --log-gates 19 --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^19', d['value'], d['kernel_ms_per_prove'])"
