#!/usr/bin/env python
"""Headline benchmark: Prover::prove wall-clock on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--log-gates G]

A step = one full prove() (V3) of a synthetic 2^G-gate circuit: 6 iNTT(n) + 6 coset-NTT(8n) +
1 coset-iNTT(8n) + 11 MSM(~n) + every O(n) pass in between, all on the GPU, producing the
1008 Proof bytes.  Wire columns, ProverKey and SRS tables are resident in HBM when the timed
region starts (they are produced by witness generation / Compiler::compile, outside prove()).
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every MSM is sharded by
SRS point range and the 192-byte partial sums are all-gathered over xGMI (strong scaling:
one proof, N GPUs).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import plonk_amd  # noqa: E402
from plonk_amd import Q  # noqa: E402

R = (1 << 256) % Q
RINV = pow(R, -1, Q)
K1, K2, K3 = 7, 13, 17
ROOT_OF_UNITY = pow(7, (Q - 1) >> 32, Q)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def synth_circuit(log_n: int, seed: int = 0x5EED0001):
    """Dense synthetic circuit of n = 2^log_n arithmetic gates (SURVEY §8d `dense` profile):
        q_M a b + q_L a + q_R b + q_O c + q_F d + q_C = 0
    with q_M, q_L, q_R, q_F, q_C uniformly random per gate (full-length selector polynomials, as
    a compiled circuit has), q_O = -1, q_arith = 1; random a_0, b_i, d_i; the output of gate i is
    wired to input a of gate i+1, so the copy permutation is non-trivial (sigma_1, sigma_3).
    Everything is kept in Montgomery representation (x~ = x R) so rows can be emitted as raw limb
    bytes without conversion.
    Returns (wires_bytes[4], key columns {name: bytes}, trivial polys {name: [ints]})."""
    import random
    n = 1 << log_n
    rnd = random.Random(seed)
    rb = rnd.getrandbits
    a = [0] * n
    b = [rb(254) % Q for _ in range(n)]
    d = [rb(254) % Q for _ in range(n)]
    qm = [rb(254) % Q for _ in range(n)]
    ql = [rb(254) % Q for _ in range(n)]
    qr = [rb(254) % Q for _ in range(n)]
    qf = [rb(254) % Q for _ in range(n)]
    qc = [rb(254) % Q for _ in range(n)]
    c = [0] * n
    cur = rb(254) % Q
    for i in range(n):
        a[i] = cur
        bi = b[i]
        # c~ = (qm~ a~ b~ R^-2 + ql~ a~ R^-1 + qr~ b~ R^-1 + qf~ d~ R^-1 + qc~)   (q_O = -1)
        cur = ((qm[i] * cur % Q * bi % Q * RINV + ql[i] * cur + qr[i] * bi + qf[i] * d[i]) % Q * RINV + qc[i]) % Q
        c[i] = cur
    tb = int.to_bytes
    wires = [b"".join(tb(x, 32, "little") for x in col) for col in (a, b, c, d)]
    omega = pow(ROOT_OF_UNITY, 1 << (32 - log_n), Q)
    t = R
    T = [0] * n
    for i in range(n):
        T[i] = t
        t = t * omega % Q
    # sigma_1[i] = K2 w^(i-1) (Output(i-1)), sigma_1[0] = w^0 ; sigma_3[i] = w^(i+1) (Left(i+1)), last = itself
    s1 = [T[0]] + [K2 * T[i - 1] % Q for i in range(1, n)]
    s3 = [T[i + 1] for i in range(n - 1)] + [K2 * T[n - 1] % Q]
    cols = {"s_sigma_1": b"".join(tb(x, 32, "little") for x in s1),
            "s_sigma_3": b"".join(tb(x, 32, "little") for x in s3)}
    for name, col in (("q_m", qm), ("q_l", ql), ("q_r", qr), ("q_f", qf), ("q_c", qc)):
        cols[name] = b"".join(tb(x, 32, "little") for x in col)
    trivial = {"q_o": [Q - 1], "q_arith": [1], "s_sigma_2": [0, K1], "s_sigma_4": [0, K3]}
    return wires, cols, trivial


def build_prover(ctx, log_n, rank, world, allgather):
    n = 1 << log_n
    srs_total = n + 7                      # the points prove() touches (commit key is trimmed to >= n + 7)
    lo, hi = plonk_amd.shard_range(srs_total, rank, world)
    tau, g = 0x5EED0000 * 0x9E3779B97F4A7C15 % Q, 0xA5A5A5A5DEADBEEF
    pts = ctx.alloc(96 * max(hi - lo, 1))
    ctx.srs_generate_dev(tau, g * pow(tau, lo, Q) % Q, hi - lo, pts.ptr)   # "random SRS": [g tau^i] G
    ctx.srs_load_dev(pts.ptr, hi - lo)
    pts.free()
    wires, cols, trivial = synth_circuit(log_n)
    polys = dict(trivial)
    buf, tmp = ctx.alloc(32 * n), ctx.alloc(32 * n)
    for name, raw in cols.items():          # selector / sigma columns -> coefficient form (compiler.rs:179-211)
        buf.upload(raw)
        ctx.ntt_dev(buf.ptr, buf.ptr, tmp.ptr, log_n, inverse=True)
        polys[name] = buf.download()
    buf.free()
    tmp.free()
    prover = plonk_amd.Prover(ctx, n, b"bench", polys, None, rank, world, srs_total, allgather)
    wbuf = ctx.alloc(4 * 32 * n)
    for k in range(4):
        wbuf.upload(wires[k], 32 * n * k)
    return prover, wbuf, srs_total


def cpu_baseline(log_n: int):
    """NTT + MSM share of one prove() on the host CPU with the C restatement (oracle/c), all
    cores, at a bounded size; reported scaled linearly to 2^log_n gates."""
    import numpy as np
    from oracle import cbind
    sample_log = min(log_n, 20)   # 2^20: every transform / MSM size is measured directly (a few seconds on 16 cores), no extrapolation
    n = 1 << sample_log
    threads = cbind.max_threads()
    rng = np.random.default_rng(1)

    def rand_fr(cnt):
        a = rng.integers(0, 256, size=(cnt, 32), dtype=np.uint8)
        a[:, 31] &= 0x3F
        return a.tobytes()
    def best_of(k, fn):
        best = 1e30
        for _ in range(k):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best
    cbind.ntt_bytes(rand_fr(1 << 12), 12, False, False, 1 << 12, threads)      # spin up the OpenMP pool
    d = rand_fr(n)
    t_ntt_n = best_of(2, lambda: cbind.ntt_bytes(d, sample_log, True, False, n, threads))
    d8 = rand_fr(n + 3)
    t_ntt_8n = best_of(2, lambda: cbind.ntt_bytes(d8, sample_log + 3, False, True, n + 3, threads))
    # distinct bases: [g tau^i]G is expensive on the CPU; tile 256 real points (timing only)
    from oracle import bls12_381 as E
    base = b"".join(E.g1_to_raw96(E.g1_mul(E.G1_GEN, 3 + 7 * i)) for i in range(256))
    pts = base * (n // 256 + 1)
    sc = rand_fr(n + 6)
    t_msm = best_of(1, lambda: cbind.msm_bytes(pts, sc, n + 6, threads))
    share_ms = (6 * t_ntt_n + 7 * t_ntt_8n + 11 * t_msm) * 1e3
    scale = float(1 << (log_n - sample_log))
    return {"value": round(share_ms * scale, 1), "unit": "ms", "cores": threads, "kind": "port",
            "sample": (f"C restatement (oracle/c, OpenMP) of the NTT+MSM share of one prove() = 6 iNTT(n) + 7 NTT(8n) + "
                       f"11 MSM(n+6), measured once at n=2^{sample_log} ({share_ms:.0f} ms: iNTT(n) {t_ntt_n * 1e3:.1f}, "
                       f"NTT(8n) {t_ntt_8n * 1e3:.1f}, MSM {t_msm * 1e3:.1f} ms) and scaled x{scale:g} linearly to "
                       f"n=2^{log_n}; excludes the O(n) passes, so it is a LOWER bound on CPU prove()")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-gates", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    allgather = None
    collective_backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # PLONK_BENCH_BACKEND=gloo + PLONK_BENCH_SHARE_GPU=1: several ranks on ONE GPU (functional test of
        # the sharded path on a 1-GPU box; RCCL itself refuses two ranks per device)
        backend = os.environ.get("PLONK_BENCH_BACKEND", "nccl")
        if os.environ.get("PLONK_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        # RCCL carries the data-path collective (CUDA tensors); gloo (CPU tensors) carries the agreement
        # below and is the fallback if RCCL cannot be brought up on this node
        dist.init_process_group(backend="cpu:gloo,cuda:nccl" if backend == "nccl" else backend, rank=rank, world_size=world)
        use_rccl = backend == "nccl"
        if use_rccl:   # self-test, then a unanimous decision so that no rank is left waiting in the other backend
            ok = 1
            try:
                probe = torch.full((8,), rank + 1, dtype=torch.uint8, device="cuda")
                got = torch.empty(8 * world, dtype=torch.uint8, device="cuda")
                dist.all_gather_into_tensor(got, probe)
                torch.cuda.synchronize()
                ok = int(got.cpu().tolist() == [r + 1 for r in range(world) for _ in range(8)])
            except Exception as e:   # noqa: BLE001
                print(f"[bench rank {rank}] RCCL self-test failed ({type(e).__name__}: {e}); falling back to gloo", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_rccl = bool(flag.item())
        collective_backend = "rccl" if use_rccl else "gloo"
        dev = "cuda" if use_rccl else "cpu"

        def allgather(send: bytes) -> bytes:   # all-gather of the MSM partial sums (RCCL over xGMI)
            t = torch.frombuffer(bytearray(send), dtype=torch.uint8).to(dev)
            out = torch.empty(world * t.numel(), dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(out, t)
            return out.cpu().numpy().tobytes()

    log_n = args.log_gates
    ctx = plonk_amd.Context(local_rank)
    t_setup = time.perf_counter()
    prover, wbuf, srs_total = build_prover(ctx, log_n, rank, world, allgather)
    t_setup = time.perf_counter() - t_setup
    blinders = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            tok = torch.zeros(1, dtype=torch.int32, device=dev)   # barrier on the backend that is known to work
            dist.all_reduce(tok)
            if dev == "cuda":
                torch.cuda.synchronize()

    proof = None
    for _ in range(args.warmup):
        proof = prover.prove_dev(wbuf.ptr, {}, blinders)
    # timed region: exactly K proofs, hipEvent pairs recorded around the dominant kernels
    ctx.profile(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = prover.prove_dev(wbuf.ptr, {}, blinders)
    barrier()
    elapsed = time.perf_counter() - t0
    acc_ms, acc_n = ctx.profile_read(1)     # msm_accumulate launches inside the timed region
    oth_ms, oth_n = ctx.profile_read(2)
    q_ms, q_n = ctx.profile_read(3)
    ctx.profile(False)
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # all ranks must have produced the identical proof
        ph = torch.frombuffer(bytearray(proof), dtype=torch.uint8).to(dev).to(torch.int32)
        ref = ph.clone()
        dist.broadcast(ref, 0)
        assert bool((ref == ph).all()), "ranks disagree on the proof bytes"
    ms_per_step = elapsed * 1e3 / args.steps
    if rank == 0:
        n = 1 << log_n
        per = (srs_total + world - 1) // world
        m_local = min(per, n + 6)          # terms of one sharded MSM on this rank
        # msm_accumulate is launched once per commitment group: (4 wires) (z) (4 quotient parts) (2 openings).
        # Algorithmic bytes of a group of b MSMs over the same bases: (32 b + 96) m  (SURVEY §8d).
        groups = (4, 1, 4, 2)
        alg_bytes_per_prove = sum((32 * b + 96) * m_local for b in groups)
        avg_acc = acc_ms / max(acc_n, 1)
        acc_ms_per_prove = acc_ms / args.steps
        achieved = alg_bytes_per_prove / (acc_ms_per_prove * 1e-3) / 1e9 if acc_ms_per_prove > 0 else 0.0
        valu_busy = None   # SQ_ACTIVE_INST_VALU / SIMD time of the same kernel, same PMC passes
        traffic = None   # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc, profiles/*/pmc.json)
        if world == 1 and log_n == 20:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc.json"))):
                try:
                    pj = json.load(open(f))
                    traffic = round(pj["traffic_bytes_per_launch"])
                    valu_busy = pj.get("valu_busy_frac")
                except Exception:
                    pass
        out = {
            "metric": "prove() wall-clock (ms) at 2^%d gates" % log_n,
            "value": round(ms_per_step, 3), "unit": "ms", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32 limbs (Fr 256-bit / Fp 384-bit Montgomery)",
            "data": "synthetic",
            "config": {"workload": "Prover::prove V3, synthetic dense 2^%d-gate arithmetic circuit, random SRS of "
                                   "n+7 points, wires/ProverKey/SRS tables resident in HBM" % log_n,
                       "gates": n, "ntt": "6 iNTT(n) + 6 cosetNTT(4n) + 1 cosetiNTT(4n) [quotient interpolated on 4n + de-aliasing; reference: 8n]" if os.environ.get("PLONK_QUOTIENT_DOMAIN", "")[:1] != "8" else "6 iNTT(n) + 6 cosetNTT(8n) + 1 cosetiNTT(8n)", "msm": "11 x ~n terms",
                       "parallelism": "msm-point-range-shard x%d" % world, "collective": collective_backend, "setup_s": round(t_setup, 1)},
            # whole-job MSM rate: all 11 x (n + 6) terms of a proof over the time rank 0 spends in its (sharded) MSM kernels
            "msm_mscalar_per_s": round(11 * (n + 6) / max((acc_ms + oth_ms) / args.steps, 1e-9) / 1e3, 2),
            "proof_blake2b": __import__("hashlib").blake2b(proof).hexdigest()[:32],
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "valu_int_fraction": None if valu_busy is None else round(valu_busy, 3), "avg_launch_ms": round(avg_acc, 4), "launches": int(acc_n),
                         "algorithmic_bytes_per_launch": alg_bytes_per_prove // len(groups),
                         "launch_groups_per_prove": list(groups),
                         "note": "integer-VALU bound (384-bit Montgomery products), not HBM bound; see DESIGN.md"},
            "kernel_ms_per_prove": {"msm_accumulate": round(acc_ms / args.steps, 3),
                                    "msm_other": round(oth_ms / args.steps, 3),
                                    "quotient_pointwise": round(q_ms / args.steps, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(log_n)
            except Exception as e:  # the baseline is a report, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "ms", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
