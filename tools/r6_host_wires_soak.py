"""Round 6: proofs from pinned HOST wire columns (plonk_prover_prove: the wire group committed column by column as the copies
land, msm_batch_device phases) against proofs from resident columns (plonk_prover_prove_dev: one grouped launch), byte for
byte, over the three bench workloads at 2^19 and 2^20 gates with several blinder sets each — the resident bytes of these
workloads are the ones tests/test_gpu_prove_sizes.py compares with the C oracle.

    python tools/r6_host_wires_soak.py [blinder sets per workload = 6]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import plonk_amd  # noqa: E402
from oracle.bls12_381 import Q  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = plonk_amd.Context(0)
t0 = time.time()
total = 0
for log_n in (19, 20):
    n = 1 << log_n
    for profile in ("dense", "bench-like", "widgets"):
        prover, wbuf, _ = bench.build_prover(ctx, log_n, 0, 1, None, profile)
        hw = [plonk_amd.PinnedBuffer(32 * n) for _ in range(4)]
        for k in range(4):
            ctx.d2h_into(hw[k].ptr, wbuf.ptr + 32 * n * k, 32 * n)
        ptrs = [b.ptr for b in hw]
        digests = set()
        for r in range(reps):
            bl = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + 977 * r + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
            resident = prover.prove_dev(wbuf.ptr, prover.public_inputs, bl)
            assert prover.describe()["wire_group_launches"] == 1
            host = prover.prove_host_ptrs(ptrs, prover.public_inputs, bl)
            assert prover.describe()["wire_group_launches"] == 3, prover.describe()
            assert host == resident, (log_n, profile, r)
            digests.add(hashlib.blake2b(host).hexdigest())
            total += 1
        assert len(digests) == reps
        for b in hw:
            b.free()
        prover.close()
        wbuf.free()
        print(f"2^{log_n} {profile}: {reps} blinder sets, host-column proofs == resident proofs", flush=True)
print(f"host_wires_soak: {total} pairs of identical proofs, {time.time() - t0:.0f} s")
