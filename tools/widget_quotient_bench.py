"""Times the quotient kernel with EVERY widget family active (GPU box).  The bench circuit gets random
q_range / q_logic / q_fixed_group_add / q_variable_group_add polynomials on top, which makes it
unsatisfied — prove() ends with CircuitUnsatisfied — but the point-wise pass runs in full and is timed
through the library's hipEvent slot 3."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import plonk_amd  # noqa: E402
from plonk_amd import Q  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    n = 1 << log_n
    ctx = plonk_amd.Context(0)
    tau, g = 0x5EED0000 * 0x9E3779B97F4A7C15 % Q, 0xA5A5A5A5DEADBEEF
    pts = ctx.alloc(96 * (n + 7))
    ctx.srs_generate_dev(tau, g, n + 7, pts.ptr)
    ctx.srs_load_dev(pts.ptr, n + 7)
    pts.free()
    wires, cols, trivial = bench.synth_circuit(log_n)
    polys = dict(trivial)
    buf, tmp = ctx.alloc(32 * n), ctx.alloc(32 * n)
    for name, raw in cols.items():
        buf.upload(raw)
        ctx.ntt_dev(buf.ptr, buf.ptr, tmp.ptr, log_n, inverse=True)
        polys[name] = buf.download()
    rng = np.random.default_rng(5)
    for name in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"):
        a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        a[:, 31] &= 0x3F
        polys[name] = a.tobytes()
    prover = plonk_amd.Prover(ctx, n, b"widget-bench", polys)
    wbuf = ctx.alloc(4 * 32 * n)
    for k in range(4):
        wbuf.upload(wires[k], 32 * n * k)
    bl = plonk_amd.fr_to_bytes_mont(list(range(1, 15)))
    ctx.profile(True)
    for it in range(4):
        if it == 1:
            ctx.profile_reset()
        try:
            prover.prove_dev(wbuf.ptr, {}, bl)
            print("unexpected: proof accepted")
        except (plonk_amd.CircuitUnsatisfied, plonk_amd.PolynomialDegreeTooLarge):
            pass
    ms, cnt = ctx.profile_read(3)
    print(f"quotient kernel with all widgets, 2^{log_n} gates: {ms / max(cnt, 1):.3f} ms per launch ({cnt} launches)")


if __name__ == "__main__":
    main()
