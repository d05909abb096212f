#!/bin/bash
# round 3, GPU call 7: affine_batch at 3 waves/SIMD; plonk_ntt_batch inside the bench process (streams of a used context)
set -u
O=gpurun_out/r3g
rm -rf $O; mkdir -p $O
timeout 600 build/ubench/affine_batch > $O/affine_batch.txt 2>&1; cat $O/affine_batch.txt
cat > /tmp/leaf2.py <<'PY'
import sys, time, ctypes, os
sys.path.insert(0, '.')
import bench, plonk_amd
ctx = plonk_amd.Context(0)
prover, wbuf, _ = bench.build_prover(ctx, 20, 0, 1, None)
bl = plonk_amd.fr_to_bytes_mont(list(range(1, 15)))
prover.prove_dev(wbuf.ptr, {}, bl)
print(bench.leaf_costs(ctx, 20))
PY
echo "== default"; timeout 200 python /tmp/leaf2.py 2>&1 | tail -1
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 200 python /tmp/leaf2.py 2>&1 | tail -1
