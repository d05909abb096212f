#!/bin/bash
# round 3, GPU call 11: which step of the bench flow makes plonk_ntt_batch lose its overlap (52 ms in bench.py, 33 ms standalone)?
set -u
cat > /tmp/leaf3.py <<'PY'
import sys, time, ctypes, os
sys.path.insert(0, '.')
import bench, plonk_amd
variant = sys.argv[1]
ctx = plonk_amd.Context(0)
prover, wbuf, _ = bench.build_prover(ctx, 20, 0, 1, None)
bl = plonk_amd.fr_to_bytes_mont(list(range(1, 15)))
proof = prover.prove_dev(wbuf.ptr, {}, bl)
n = 1 << 20
if 'P' in variant:
    ctx.profile(True); ctx.profile_reset(); prover.prove_dev(wbuf.ptr, {}, bl); ctx.profile_read(1); ctx.profile(False)
if 'H' in variant:
    hw = [plonk_amd.PinnedBuffer(32 * n) for _ in range(4)]
    for k in range(4): ctx.d2h_into(hw[k].ptr, wbuf.ptr + 32 * n * k, 32 * n)
    assert prover.prove_host_ptrs([b.ptr for b in hw], {}, bl) == proof
    for b in hw: b.free()
if 'C' in variant:
    prover.close(); wbuf.free()
if 'R' in variant:
    bench.ntt_roofline(ctx, 20, False)
r = bench.leaf_costs(ctx, 20)
print(variant, r['plonk_ntt_batch5_2p23_ms'], r['plonk_ntt_batch5_2p23_coset_evaluations_ms'])
PY
for v in none P H C R PHCR; do timeout 200 python /tmp/leaf3.py $v 2>&1 | tail -1; done
