#!/bin/bash
# round 4, GPU call 13: sharded grand product: multirank parity (incl. 2^20 for W = 2, 8), rank-alone A/B
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4m
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m "gpu" -k "grand_product or 2p20 or sharded_quotient" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
cd /tmp
for V in 0 1; do
  PLONK_SHARD_Z=$V python $R/tools/rank_alone.py 20 6 8 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^20 W=8 shard_z=$V', d['prove_ms_rank_alone'], d['kernel_ms'])"
  PLONK_SHARD_Z=$V python $R/tools/rank_alone.py 20 6 2 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^20 W=2 shard_z=$V', d['prove_ms_rank_alone'], d['kernel_ms'])"
  PLONK_SHARD_Z=$V python $R/tools/rank_alone.py 22 3 8 2> $O/ra.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^22 W=8 shard_z=$V', d['prove_ms_rank_alone'], d['kernel_ms'])"
done
