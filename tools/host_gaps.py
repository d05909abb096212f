"""Host time on the critical path of prove() (slots 8-10 of plonk_profile_read, prover.hip HostGap): how long the device's
main stream waits for the host at the five transcript points of a proof.
usage: python tools/host_gaps.py [log_gates ...]   ->  one JSON line per size: ms per proof"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import plonk_amd  # noqa: E402
from oracle.bls12_381 import Q  # noqa: E402


def run(log_n, steps):
    ctx = plonk_amd.Context(0)
    blinders = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
    prover, wbuf, _ = bench.build_prover(ctx, log_n, 0, 1, None, os.environ.get("HG_PROFILE", "dense"))
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
    prof = (time.perf_counter() - t0) * 1e3 / steps
    out = {"log_gates": log_n, "profile": os.environ.get("HG_PROFILE", "dense"), "lib": os.path.basename(os.environ.get("PLONK_HIP_LIB", "default")), "prove_ms": round(plain, 3), "prove_ms_profiled": round(prof, 3)}
    for name, slot in (("host_finish_commitments", 8), ("host_gap_sync_to_next_launch", 9), ("host_blocked_in_sync", 10)):
        total, cnt = ctx.profile_read(slot)
        out[name + "_ms"] = round(total / steps, 4)
        out[name + "_n"] = cnt / steps
    ctx.profile(False)
    prover.close()
    wbuf.free()
    ctx.close()
    return out


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [12, 16, 20]
    for lg in sizes:
        print(json.dumps(run(lg, int(os.environ.get("HG_STEPS", 20 if lg <= 18 else 5)))), flush=True)
