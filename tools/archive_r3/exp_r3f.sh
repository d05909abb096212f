#!/bin/bash
# round 3, GPU call 6: (a) shared-inversion affine additions vs the XYZZ mixed addition (tools/ubench/affine_batch.hip);
# (b) does plonk_ntt_batch's pipeline overlap once the runtime has more hardware queues than streams?  Output: gpurun_out/r3f/
set -u
O=gpurun_out/r3f
rm -rf $O; mkdir -p $O
timeout 600 build/ubench/affine_batch > $O/affine_batch.txt 2>&1; cat $O/affine_batch.txt
cat > /tmp/leaf.py <<'PY'
import sys, time, ctypes
sys.path.insert(0, '.')
import plonk_amd
ctx = plonk_amd.Context(0)
L = 23; n = 1 << L
bufs = [plonk_amd.PinnedBuffer(32 * n) for _ in range(5)]
for b in bufs: b.write(bytes(32 * n))
vp = ctypes.c_void_p
arr = (vp * 5)(*[vp(b.ptr) for b in bufs])
for lens in (None, (ctypes.c_uint64 * 5)(*[(n >> 3) + 3] * 5)):
    ctx._check(ctx.lib.plonk_ntt_batch(ctx.handle, arr, 5, L, 0, 1, lens))
    t0 = time.perf_counter()
    ctx._check(ctx.lib.plonk_ntt_batch(ctx.handle, arr, 5, L, 0, 1, lens))
    print("plonk_ntt_batch x5 2^23", "full input" if lens is None else "n+3 coefficients in", round((time.perf_counter() - t0) * 1e3, 2), "ms")
ctx._check(ctx.lib.plonk_ntt(ctx.handle, vp(bufs[0].ptr), L, 0, 1, n))
t0 = time.perf_counter()
ctx._check(ctx.lib.plonk_ntt(ctx.handle, vp(bufs[0].ptr), L, 0, 1, n))
print("plonk_ntt 2^23", round((time.perf_counter() - t0) * 1e3, 2), "ms")
PY
echo "== default queues"; timeout 120 python /tmp/leaf.py 2>&1 | tail -3
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 120 python /tmp/leaf.py 2>&1 | tail -3
echo "== GPU_MAX_HW_QUEUES=16"; GPU_MAX_HW_QUEUES=16 timeout 120 python /tmp/leaf.py 2>&1 | tail -3
