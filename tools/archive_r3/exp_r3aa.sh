#!/bin/bash
# round 3, GPU call 27: NTT passes with batched loads (branch-free load phase, per-mode epilogue loops) — tests, standalone passes, bench
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3aa
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu > $O/ntt_tests.log 2>&1; echo "ntt tests rc=$?"; tail -3 $O/ntt_tests.log
for M in 0 1; do
  PLONK_NTT_DIRECT=$M python tools/ntt_passes.py 20 10 > $O/plain$M.txt 2>&1; echo "direct=$M"; cat $O/plain$M.txt
done
python tools/ntt_passes.py 22 5 2>&1 | tee $O/plain_22.txt
python tools/ntt_passes.py 16 20 2>&1 | tee $O/plain_16.txt
for T in a b; do
for M in 0 1; do
  PLONK_NTT_DIRECT=$M timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/b$M$T.json 2> $O/b$M$T.err
  python - <<PY
import json
j = json.loads(open('$O/b$M$T.json').read().strip().splitlines()[-1])
print('direct=$M', j['value'], j.get('kernel_ms_per_prove'), j.get('proof_blake2b'))
PY
done
done
