#!/usr/bin/env python
"""Peer failure inside a sharded proof, on ONE GPU, through the stand-in transport (test infrastructure).

  python tests/fake_rccl/peer_death.py WORLD LOG_GATES DIE_RANK DIE_AT [noabort]       e.g.  2 12 1 alltoall:1

Starts WORLD ranks that share device 0.  Every rank brings up a communicator on libfakerccl.so, runs the library's
self-test (all-gather #1, all-to-all #1 of the transport), builds its shard of the bench prover and proves.  The
stand-in makes rank DIE_RANK _exit(17) when it enters the collective DIE_AT names, counted from the start of the first
proof (`alltoall:1` = its quotient all-to-all, `allgather:2` = the all-gather of the z commitment), i.e. in the middle of a proof whose other ranks are already inside — or about to enter —
the same collective.  The survivors must come back from plonk_prover_prove_dev with PLONK_ERR_STATE after about
PLONK_COMM_TIMEOUT_MS (comm.hip comm_sync: poll, ncclCommAbort, bounded drain) instead of hanging.

The launcher prints one JSON line: {"ranks": [{"rank", "exit", "rc", "seconds", "error"} ...]}.
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
TIMEOUT_MS = 4000


def rank_main(rank: int, world: int, log_n: int, uid_path: str) -> int:
    sys.path.insert(0, ROOT)
    os.environ["PLONK_BENCH_SHARE_GPU"] = "1"
    import bench
    import plonk_amd
    plonk_amd.Context.comm_set_library(FAKE)
    if rank == 0:
        uid = plonk_amd.Context.comm_unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                return 5
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    ctx = plonk_amd.Context(0)
    ctx.comm_init(uid, rank, world)
    ctx.comm_selftest()
    # the fault is armed only AFTER the prover exists: FAKE_RCCL_DIE_AT then counts the collectives of the proof (all-gather 1 =
    # the wire commitments, 2 = z; all-to-all 1 = the quotient), however many the set-up took
    die_at = os.environ.pop("PEER_DEATH_AT")
    prover, wbuf, _ = bench.build_prover(ctx, log_n, rank, world, None, "dense")
    # the SHORT time-out only from here on: set-up collectives (prover creation has four since round 6) see the ranks seconds
    # apart on a busy box — what is under test is a peer that dies INSIDE a proof.  The self-test is the barrier in front of it.
    ctx.comm_selftest()
    g = ctx.get_config()
    g.comm_timeout_ms = TIMEOUT_MS
    ctx.set_config(g)
    import ctypes
    ctypes.CDLL(FAKE).fake_rccl_arm()
    os.environ["FAKE_RCCL_DIE_AT"] = die_at
    blinders = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % plonk_amd.Q for i in range(14)])
    t0 = time.perf_counter()
    out = {"rank": rank, "rc": 0, "error": ""}
    try:
        prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
    except plonk_amd.PlonkError as e:
        out["rc"], out["error"] = e.code, str(e)[:300]
    out["seconds"] = round(time.perf_counter() - t0, 2)
    # a second proof on the context that lost its communicator must be refused at once, not hang
    t0 = time.perf_counter()
    try:
        prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
        out["rc_second"] = 0
    except plonk_amd.PlonkError as e:
        out["rc_second"] = e.code
    out["seconds_second"] = round(time.perf_counter() - t0, 2)
    if os.environ.get("FAKE_RCCL_ABORT_FAILS") == "1":
        # the abort did not release the collective's kernel: the context is POISONED (comm.hip comm_sync).  Every entry point
        # that would queue behind the dead kernel must refuse at once, and the destroy calls must come back after their
        # bounded poll (2 s each) instead of sitting in hipStreamSynchronize / hipFree — ADVICE r5
        for name, call in (("msm", lambda: ctx.msm([1, 2, 3])), ("ntt", lambda: ctx.ntt([1, 2, 3, 4], 2)),
                           ("sync", ctx.sync), ("h2d", lambda: ctx.alloc(64).upload(bytes(64)))):
            t0 = time.perf_counter()
            try:
                call()
                out["rc_" + name] = 0
            except plonk_amd.PlonkError as e:
                out["rc_" + name] = e.code
            out["seconds_" + name] = round(time.perf_counter() - t0, 2)
        t0 = time.perf_counter()
        prover.close()
        wbuf = None
        ctx.close()
        out["seconds_teardown"] = round(time.perf_counter() - t0, 2)
    print("RANKJSON " + json.dumps(out), flush=True)
    os._exit(0)   # no teardown through a communicator whose peer is gone


def main() -> int:
    if "PEER_DEATH_RANK" in os.environ:
        return rank_main(int(os.environ["PEER_DEATH_RANK"]), int(sys.argv[1]), int(sys.argv[2]), os.environ["PEER_DEATH_UID"])
    world, log_n, die_rank, die_at = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    no_abort = len(sys.argv) > 5 and sys.argv[5] == "noabort"   # the transport's abort fails: the survivor's context ends up poisoned
    with tempfile.TemporaryDirectory() as td:
        uid_path = os.path.join(td, "uid")
        procs = []
        for r in range(world):
            env = dict(os.environ, PEER_DEATH_RANK=str(r), PEER_DEATH_UID=uid_path, HSA_ENABLE_IPC_MODE_LEGACY="0",
                       FAKE_RCCL_DIE_RANK=str(die_rank), PEER_DEATH_AT=die_at,
                       FAKE_RCCL_KERNEL_TIMEOUT_S="25" if no_abort else "60", FAKE_RCCL_ABORT_FAILS="1" if no_abort else "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:3]], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        ranks = []
        for r, p in enumerate(procs):
            try:
                so, se = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                p.kill()
                so, se = p.communicate()
                ranks.append({"rank": r, "exit": "timeout", "stderr": se[-600:]})
                continue
            rec = {"rank": r, "exit": p.returncode}
            for line in so.splitlines():
                if line.startswith("RANKJSON "):
                    rec.update(json.loads(line[9:]))
            if p.returncode not in (0, 17):
                rec["stderr"] = se[-600:]
            ranks.append(rec)
    print(json.dumps({"world": world, "log_gates": log_n, "die_rank": die_rank, "die_at": die_at,
                      "timeout_ms": TIMEOUT_MS, "ranks": ranks}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
