"""Kernel micro-benchmarks on one GPU (device-resident operands).  Prints one line per
case; used with rocprofv3 to produce profiles/."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd  # noqa: E402


def rand_fr_bytes(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x3F            # < 2^254 < q : valid Montgomery residues
    return a.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntt", type=int, nargs="*", default=[16, 20, 23])
    ap.add_argument("--msm", type=int, nargs="*", default=[16, 20])
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    ctx = plonk_amd.Context(0)
    for L in args.ntt:
        N = 1 << L
        src, dst, tmp = ctx.alloc(32 * N), ctx.alloc(32 * N), ctx.alloc(32 * N)
        src.upload(rand_fr_bytes(N, L))
        for mode, (inv, coset, in_len) in {"fwd": (False, False, N), "inv": (True, False, N),
                                           "coset_fwd_n/8": (False, True, N // 8 + 3),
                                           "coset_inv": (True, True, N)}.items():
            ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, in_len)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, in_len)
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3 / args.iters
            print(f"ntt L={L} {mode:14s} {ms:9.3f} ms  {N / ms / 1e3:9.1f} Melem/s  "
                  f"algGB/s={64 * N / ms / 1e6:8.1f}", flush=True)
        for b in (src, dst, tmp):
            b.free()
    for lm in args.msm:
        n = (1 << lm) + 7
        m = (1 << lm) + 6
        pts = ctx.alloc(96 * n)
        t0 = time.perf_counter()
        ctx.srs_generate_dev(0x1234567, 0x7654321, n, pts.ptr)
        t1 = time.perf_counter()
        ctx.srs_load_dev(pts.ptr, n)
        t2 = time.perf_counter()
        print(f"srs gen n={n} {1e3 * (t1 - t0):.1f} ms; table build {1e3 * (t2 - t1):.1f} ms", flush=True)
        pts.free()
        sc = ctx.alloc(32 * m)
        sc.upload(rand_fr_bytes(m, 1000 + lm))
        out = ctx.alloc(128)
        ctx.msm_dev(sc.ptr, m, out.ptr)
        ctx.sync()
        ctx.profile(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            ctx.msm_dev(sc.ptr, m, out.ptr)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / args.iters
        acc_ms, acc_n = ctx.profile_read(1)
        oth_ms, _ = ctx.profile_read(2)
        ctx.profile(False)
        print(f"msm m={m} {ms:9.3f} ms  {m / ms / 1e3:8.2f} Mscalar/s  accumulate={acc_ms / max(acc_n, 1):.3f} ms "
              f"other={oth_ms / max(acc_n, 1):.3f} ms algGB/s={128 * m / ms / 1e6:.1f}", flush=True)
        sc.free()
        out.free()


if __name__ == "__main__":
    main()
