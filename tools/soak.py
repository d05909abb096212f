"""Soak test (GPU box): the same proof N times in a row must come out byte-identical every time —
catches rare stream-ordering hazards (side-stream transforms vs. main-stream kernels, buffer reuse
across proofs) that a single run would miss.  Usage: python tools/soak.py [log_gates] [iterations]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import plonk_amd  # noqa: E402
from plonk_amd import Q  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ctx = plonk_amd.Context(0)
    prover, wbuf, _ = bench.build_prover(ctx, log_n, 0, 1, None)
    bl = [plonk_amd.fr_to_bytes_mont([(0xB11D0000 + 17 * k + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)]) for k in range(3)]
    ref = [hashlib.blake2b(prover.prove_dev(wbuf.ptr, {}, b)).hexdigest() for b in bl]
    t0 = time.time()
    bad = 0
    for it in range(iters):
        k = it % 3                                  # alternate blinders so that consecutive proofs differ
        h = hashlib.blake2b(prover.prove_dev(wbuf.ptr, {}, bl[k])).hexdigest()
        if h != ref[k]:
            bad += 1
            print(f"MISMATCH at iteration {it}: {h} != {ref[k]}", flush=True)
    print(f"soak 2^{log_n}: {iters} proofs in {time.time() - t0:.1f} s, mismatches: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
