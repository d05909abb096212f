"""ONE rank of a W-rank sharded proof timed alone on one GPU (measure the per-rank critical path instead of
modelling it).  plonk_comm_measure_loopback(ctx, 1) makes every collective return the rank's own contribution in its peers' places with
local copies, so the rank runs exactly its kernels over its share of the points and coefficients and the same host sequence;
what is NOT in the number is the transport itself (6 small all-gathers + one all-to-all per proof over xGMI) and waiting for
slower peers.  The proofs are wrong by construction: prove() must end in PLONK_ERR_UNSAT at its final identity check.

usage: python tools/rank_alone.py [log_gates] [steps] [worlds, e.g. 2,8]  ->  one line per (W, rank) as JSON"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import plonk_amd  # noqa: E402
from oracle.bls12_381 import Q  # noqa: E402


def run(log_n, world, rank, steps):
    ctx = plonk_amd.Context(0)
    ctx.comm_measure_loopback(True)
    blinders = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
    calls = [0]

    def never(send):   # the library must not reach the callback in loop-back mode
        calls[0] += 1
        return send * world

    prover, wbuf, srs_total = bench.build_prover(ctx, log_n, rank, world, never if world > 1 else None)

    def prove():
        try:
            prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
            return world == 1
        except plonk_amd.CircuitUnsatisfied:         # PLONK_ERR_UNSAT: the final identity check, after all the work
            assert world > 1
            return True

    assert prove() and prove()
    ctx.sync()
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        prove()
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    slots = {}
    for name, slot in (("ntt", 0), ("msm_accumulate", 1), ("msm_other", 2), ("quotient_pointwise", 3), ("rounds_1_2_polynomials", 4)):
        total, _ = ctx.profile_read(slot)
        slots[name] = round(total / steps, 3)
    host = {}   # host time on the rank's critical path (slots 8-10, prover.hip HostGap): ms per proof, synchronisations per proof
    for name, slot in (("finish_commitments", 8), ("sync_to_next_launch", 9), ("blocked_in_sync", 10)):
        total, cnt = ctx.profile_read(slot)
        host[name + "_ms"] = round(total / steps, 4)
        host[name + "_n"] = cnt / steps
    lo, hi = plonk_amd.shard_range(srs_total, rank, world)
    out = {"log_gates": log_n, "world": world, "rank": rank, "points": hi - lo, "prove_ms_rank_alone": round(ms, 3), "kernel_ms": slots, "host_ms": host,
           "table_rows": ctx.table_rows() if hasattr(ctx, "table_rows") else None, "callback_calls": calls[0],
           "wire_split": getattr(bench.build_prover, "wire_split", "range") if world > 1 else None,   # PLONK_BENCH_WIRE_SPLIT=commitment
           "lagrange_points": prover.describe()["lagrange_points"],
           # what the commit-key groups of this rank ran as (the last group of a proof is the two opening witnesses over the commit key)
           "commit_key_plan": {k: v for k, v in ctx.last_msm().items() if k in ("table_rows", "bucket_bits", "digit_width", "slice_entries", "accumulate_kernel")},
           "forced": {k: v for k, v in os.environ.items() if k.startswith("PLONK_MSM_") or k == "PLONK_BENCH_WIRE_SPLIT"}}
    prover.close()
    wbuf.free()
    ctx.close()
    return out


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    worlds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
    for world in worlds:
        for rank in sorted({0, world - 1}) if len(sys.argv) <= 3 else [0]:
            print(json.dumps(run(log_n, world, rank, steps)), flush=True)


if __name__ == "__main__":
    main()
