"""Kernel time of every phase of the MSM pipeline inside prove(), measured with hipEvents between the launches
(PLONK_PROF_FINE=1) — no profiler attached: rocprofv3's serialised dispatches inflate some of these kernels 2-3x
(profiles/r04a vs r04b: msm_bucket_sum 400 us traced, ~150 us between events).
usage: python tools/msm_phases.py [log_gates ...]   ->  one JSON line per size: ms per proof"""
import json
import os
import sys
import time

os.environ["PLONK_PROF_FINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import plonk_amd  # noqa: E402
from oracle.bls12_381 import Q  # noqa: E402

PHASES = ["sort", "accumulate", "bucket_sums", "heavy", "rowcol", "bits"]


def run(log_n, steps):
    ctx = plonk_amd.Context(0)
    blinders = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
    prover, wbuf, _ = bench.build_prover(ctx, log_n, 0, 1, None)
    for _ in range(3):
        prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
    ctx.sync()
    plain = (time.perf_counter() - t0) * 1e3 / steps
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    out = {"log_gates": log_n, "prove_ms": round(plain, 3), "prove_ms_with_events": round(ms, 3)}
    for base, tag in ((16, "groups_of_4"), (24, "groups_of_1_2")):
        d = {}
        for i, ph in enumerate(PHASES):
            total, n = ctx.profile_read(base + i)
            d[ph] = round(total / steps, 4)
        d["launches_per_proof"] = n / steps
        out[tag] = d
    for name, slot in (("msm_accumulate", 1), ("msm_other", 2), ("quotient", 3), ("rounds_1_2", 4)):
        total, _ = ctx.profile_read(slot)
        out[name] = round(total / steps, 3)
    prover.close()
    wbuf.free()
    ctx.close()
    return out


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [16]
    for lg in sizes:
        print(json.dumps(run(lg, 20 if lg <= 18 else 5)), flush=True)
