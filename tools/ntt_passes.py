"""Standalone device-resident transforms for profiling the NTT passes: `rocprofv3 --kernel-trace --stats -- python
tools/ntt_passes.py [log_n]` gives per-pass kernel durations without the prover's concurrent MSM kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ctx = plonk_amd.Context()
    n = 1 << log_n
    for name, L, inv, coset, in_len in (("coset_ntt_4n", log_n + 2, False, True, n + 3), ("coset_intt_4n", log_n + 2, True, True, 4 * n),
                                        ("intt_n", log_n, True, False, n), ("ntt_n", log_n, False, False, n)):
        N = 1 << L
        src, dst, tmp = ctx.alloc(32 * N), ctx.alloc(32 * N), ctx.alloc(32 * N)
        src.upload(os.urandom(32 * (1 << 16)))
        ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, in_len)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, in_len)
        ctx.sync()
        print(f"{name} 2^{L}: {(time.perf_counter() - t0) * 1e3 / iters:.4f} ms", flush=True)
        for b in (src, dst, tmp):
            b.free()


if __name__ == "__main__":
    main()
