#!/usr/bin/env python
"""Headline benchmark: Prover::prove wall-clock on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--log-gates G] [--profile dense|bench-like|widgets]

A step = one full prove() (V3) of a synthetic 2^G-gate circuit: 5 iNTT(n) (a, b, c, d, z; +1 for the
public-input polynomial when the circuit has public inputs) + 5 coset-NTT(4n) (+1) + 1 coset-iNTT(4n)
+ 11 MSM(~n) + every O(n) pass in between, all on the GPU, producing the 1008 Proof bytes.  Wire
columns, ProverKey and SRS tables are resident in HBM when the timed region starts (they are produced
by witness generation / Compiler::compile, outside prove()).

N > 1: one process per GPU.  The data path is RCCL INSIDE the library (plonk_comm_init: ncclAllGather /
ncclAllToAll on the library's stream over xGMI) — torch.distributed (gloo) is only the out-of-band
channel that hands the ncclUniqueId to the ranks and runs the barriers around the timed region.
Every MSM is sharded by SRS point range, the quotient by residue class of the coset, evaluations /
linearisation / openings by coefficient range (strong scaling: one proof, N GPUs).  If RCCL cannot be
brought up (e.g. several ranks sharing one device in the tests) all ranks agree on the host-callback
transport over gloo.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver

import bench_circuits as BC  # noqa: E402
import plonk_amd  # noqa: E402
from plonk_amd import Q  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
TAU, G_SCALAR = 0x5EED0000 * 0x9E3779B97F4A7C15 % Q, 0xA5A5A5A5DEADBEEF   # "random SRS" [g tau^i] G


def synth_circuit(log_n: int):
    """the headline `dense` workload (kept under this name for the tests)"""
    return BC.arithmetic_circuit(log_n, "dense")


def circuit_polys(ctx, log_n, profile):
    """(wires[4] bytes, {name: coefficient-form polynomial}, public inputs) of a profile; the selector /
    sigma columns are interpolated on the GPU (compiler.rs:179-211)."""
    n = 1 << log_n
    if profile == "widgets":
        wires, cols, pi = BC.widget_circuit(log_n)
        polys = {}
    else:
        wires, cols, trivial = BC.arithmetic_circuit(log_n, profile, workers=8 if log_n > 20 else 0)
        polys, pi = dict(trivial), {}
    circuit_polys.q_m_column = cols.get("q_m")   # one evaluation-form column, for the tests that pin the shared inputs
    buf, tmp = ctx.alloc(32 * n), ctx.alloc(32 * n)
    for name, raw in cols.items():
        buf.upload(raw)
        ctx.ntt_dev(buf.ptr, buf.ptr, tmp.ptr, log_n, inverse=True)
        polys[name] = buf.download()
    buf.free()
    tmp.free()
    return wires, polys, pi


def build_prover(ctx, log_n, rank, world, allgather, profile="dense", from_circuit=False, keep_inputs=False, on_inputs=None):
    """keep_inputs: leave (wires, coefficient-form polynomials, public inputs) in build_prover.inputs — the tests hand them to
    the C oracle for the byte comparison; on_inputs(inputs): called with the same tuple as soon as it exists, BEFORE the
    prover is built (the 2^22 test starts its CPU oracle there, beside the GPU's set-up)"""
    n = 1 << log_n
    srs_total = n + 7                      # the points prove() touches (commit key is trimmed to >= n + 7)
    lo, hi = plonk_amd.shard_range(srs_total, rank, world)
    lag_slice = None
    eff = ctx.get_config()
    if world in (2, 4, 8) and n >= 64 and not eff.wire_commit and eff.shard_quotient >= 0:
        # multi-GPU: the Lagrange-basis key needs the WHOLE commit key once (a group FFT); every rank derives it from the
        # full synthetic key and keeps its slice (a real deployment computes it once and ships the slices)
        def whole_key():
            full = ctx.alloc(96 * (n + 2))
            ctx.srs_generate_dev(TAU, G_SCALAR, n + 2, full.ptr)
            # this load only feeds the group FFT (row 0 of the tables): the 16 window rows are enough, the 256 bit-position
            # rows of the full key would be 32 GiB per rank of pure setup
            cfg = ctx.get_config()
            saved = cfg.table_mode
            cfg.table_mode = plonk_amd.TABLE_WINDOW
            ctx.set_config(cfg)
            try:
                ctx.srs_load_dev(full.ptr, n + 2)
            finally:
                cfg.table_mode = saved
                ctx.set_config(cfg)
            full.free()
            return ctx.lagrange_key(log_n)
        # (test sessions only — PLONK_CIRCUIT_CACHE: ranks that SHARE one GPU would each run the same group FFT one after the
        # other, 8 x 3.5 s at 2^22 gates; the key of (TAU, G_SCALAR, size) is computed by the first rank and read by the rest)
        key = BC._cached("lagrange_key", log_n, "tau5eed", whole_key)
        llo, lhi = min(lo, n + 2), min(hi, n + 2)
        # 2 / 4 ranks: every rank takes the WHOLE Lagrange-basis key and commits to whole wire columns instead of a point range
        # of every column (prover.hip lag_whole; DESIGN.md section 5: rank alone at 2^20, same box, 18.34 -> 17.98 ms for W = 2,
        # 9.08 -> 8.68 for W = 4; 32 GiB of table per rank).  PLONK_BENCH_WIRE_SPLIT=range restores the split by point range.
        # (tests only: a comma-separated list gives every rank its own split — ranks that DISAGREE must be refused by the library)
        split = os.environ.get("PLONK_BENCH_WIRE_SPLIT", "commitment").split(",")
        build_prover.wire_split = "commitment" if (world in (2, 4) and split[rank % len(split)] == "commitment") else "range"
        lag_slice = key if build_prover.wire_split == "commitment" else key[96 * llo:96 * lhi]
        if os.environ.get("PLONK_BENCH_CORRUPT_LAGRANGE") == str(rank) and len(lag_slice) >= 192:
            # (tests only) two points of this rank's key swapped: every point is still on the curve and in the subgroup, but the
            # key is no longer the Lagrange-basis form of the commit key — the check at prover creation must say so
            lag_slice = lag_slice[96:192] + lag_slice[:96] + lag_slice[192:]
        del key
    # the rank's slice of the commit key is produced once (on the device: the reference's setup is O(n * 255)
    # group operations), parked in PINNED HOST memory like a key read from disk, and then STREAMED into the
    # context by plonk_srs_load (double-buffered 2^18-point chunks; BASELINE config 5)
    pts = ctx.alloc(96 * max(hi - lo, 1))
    ctx.srs_generate_dev(TAU, G_SCALAR * pow(TAU, lo, Q) % Q, hi - lo, pts.ptr)
    host = plonk_amd.PinnedBuffer(96 * max(hi - lo, 1))
    ctx.d2h_into(host.ptr, pts.ptr, 96 * (hi - lo))
    pts.free()
    t0 = time.perf_counter()
    ctx.srs_load_host_ptr(host.ptr, hi - lo)
    build_prover.srs_stream_s = time.perf_counter() - t0
    host.free()
    build_prover.witness_values = None
    if from_circuit:
        # Compiler::preprocess on the device (plonk_compile): gate columns + witness indices in, prover + VerifierKey out
        cc = BC.widget_columns(log_n) if profile == "widgets" else BC.arithmetic_columns(log_n, profile, workers=8 if log_n > 20 else 0)
        wires, pi = cc["columns"], cc["public_inputs"]
        t0 = time.perf_counter()
        prover = plonk_amd.Prover.compile(ctx, b"bench", cc["selectors"], cc["wires"], cc["witnesses"], rank, world, srs_total,
                                          allgather, lag_slice)
        build_prover.compile_s = time.perf_counter() - t0
        build_prover.witness_values = cc["values"]
    else:
        wires, polys, pi = circuit_polys(ctx, log_n, profile)
        build_prover.selectors_nonzero = sum(1 for k in ("q_m", "q_l", "q_r", "q_o", "q_f")
                                             if isinstance(polys.get(k), (bytes, bytearray)) and polys[k].strip(b"\0")
                                             or (not isinstance(polys.get(k), (bytes, bytearray)) and polys.get(k) and any(polys[k])))
        build_prover.inputs = (wires, polys, pi, circuit_polys.q_m_column) if keep_inputs else None
        if on_inputs is not None:
            on_inputs((wires, polys, pi, circuit_polys.q_m_column))
        circuit_polys.q_m_column = None
        t0 = time.perf_counter()
        prover = plonk_amd.Prover(ctx, n, b"bench", polys, None, rank, world, srs_total, allgather, lag_slice)
        build_prover.create_s = time.perf_counter() - t0
    wbuf = ctx.alloc(4 * 32 * n)
    for k in range(4):
        wbuf.upload(wires[k], 32 * n * k)
    prover.public_inputs = pi
    return prover, wbuf, srs_total


def host_wire_legs(ctx, prover, wbuf, n, blinders, proof, steps):
    """prove() with the four wire columns handed over as pinned HOST buffers (what a shim around Prover::prove holds,
    prover.rs:446-460): PCIe-inclusive, never `value`.  Returns ms per proof; the bytes must be the resident proof's."""
    hw = [plonk_amd.PinnedBuffer(32 * n) for _ in range(4)]
    try:
        for k in range(4):
            ctx.d2h_into(hw[k].ptr, wbuf.ptr + 32 * n * k, 32 * n)
        ptrs = [b.ptr for b in hw]
        assert prover.prove_host_ptrs(ptrs, prover.public_inputs, blinders) == proof
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(steps):
            prover.prove_host_ptrs(ptrs, prover.public_inputs, blinders)
        ctx.sync()
        ms = round((time.perf_counter() - t1) * 1e3 / steps, 3)
        # the floor of this path: no kernel of prove() can start before the FIRST column has crossed PCIe (every later column
        # travels under the commitment work of the ones before it, prover.hip by_column) — measured here, same buffers
        ctx.h2d_from(wbuf.ptr, hw[0].ptr, 32 * n)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            ctx.h2d_from(wbuf.ptr, hw[0].ptr, 32 * n)   # (column a onto itself: the same bytes)
            ctx.sync()
        host_wire_legs.first_column_h2d_ms = round((time.perf_counter() - t1) * 1e3 / 5, 3)
        return ms
    finally:
        for b in hw:
            b.free()


def time_profile(ctx, log_n, profile, steps, blinders, digest=None, host_legs=None, from_circuit=False):
    """ms per prove() of another workload on this (single) GPU; digest: a dict that receives the proof's blake2b;
    host_legs: a dict that receives the PCIe-inclusive legs of the same prover (pinned host wire columns)"""
    prover, wbuf, _ = build_prover(ctx, log_n, 0, 1, None, profile, from_circuit=from_circuit)
    try:   # (a failure must not leave a 2^22 prover's tens of GB behind: the caller may fall back to another constructor)
        proof = prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
        if digest is not None:
            digest["proof_blake2b"] = hashlib.blake2b(proof).hexdigest()[:32]
            digest["prover_built_by"] = "plonk_compile (gate columns)" if from_circuit else "plonk_prover_create (coefficient forms)"
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        if host_legs is not None:
            try:
                host_legs["prove_ms_host_wires_pinned"] = host_wire_legs(ctx, prover, wbuf, 1 << log_n, blinders, proof, steps)
                host_legs["first_column_h2d_ms"] = host_wire_legs.first_column_h2d_ms
                host_legs["pcie_floor_ms"] = round(ms + host_wire_legs.first_column_h2d_ms, 3)   # resident proof + the one copy nothing can hide
                host_legs["wire_bytes_mib"] = 4 * 32 << (log_n - 20)
                host_legs["prove_ms_resident"] = round(ms, 3)
                if from_circuit and build_prover.witness_values is not None:
                    # a proof that starts from the witness TABLE in pinned host memory (plonk_prover_prove_witnesses: one Fr per
                    # witness up, the four padded columns gathered on the device, prover.rs:446-460)
                    vals = build_prover.witness_values
                    host = plonk_amd.PinnedBuffer(len(vals))
                    try:
                        host.write(vals)
                        cnt = len(vals) // 32
                        assert prover.prove_witnesses_ptr(host.ptr, cnt, prover.public_inputs, blinders) == proof
                        ctx.sync()
                        t1 = time.perf_counter()
                        for _ in range(steps):
                            prover.prove_witnesses_ptr(host.ptr, cnt, prover.public_inputs, blinders)
                        ctx.sync()
                        host_legs["prove_ms_from_witness_table_pinned"] = round((time.perf_counter() - t1) * 1e3 / steps, 3)
                        host_legs["witness_table_mib"] = round(len(vals) / 2**20, 1)
                    finally:
                        host.free()
            except Exception as e:   # noqa: BLE001
                host_legs["error"] = repr(e)
    finally:
        prover.close()
        wbuf.free()
    return round(ms, 3)


def compile_costs(ctx, log_n, blinders):
    """Compiler::preprocess on the device (plonk_compile, `dense` gate columns from host memory) beside the
    coefficient-form constructor, and a proof that starts from the witness table in pinned host memory."""
    n = 1 << log_n
    pts = ctx.alloc(96 * (n + 7))
    ctx.srs_generate_dev(TAU, G_SCALAR, n + 7, pts.ptr)
    ctx.srs_load_dev(pts.ptr, n + 7)
    pts.free()
    cc = BC.arithmetic_columns(log_n, "dense")
    ctx.sync()
    t0 = time.perf_counter()
    prover = plonk_amd.Prover.compile(ctx, b"bench", cc["selectors"], cc["wires"], cc["witnesses"])
    t_compile = time.perf_counter() - t0
    host = plonk_amd.PinnedBuffer(len(cc["values"]))
    host.write(cc["values"])
    proof = prover.prove_witnesses_ptr(host.ptr, cc["witnesses"], {}, blinders)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        prover.prove_witnesses_ptr(host.ptr, cc["witnesses"], {}, blinders)
    ctx.sync()
    t_prove = (time.perf_counter() - t0) / 3
    vk = prover.vk_commitments()
    prover.close()
    host.free()
    return {"gates": n, "plonk_compile_ms": round(t_compile * 1e3, 1),
            "of_which": "11 selector + 4 sigma interpolations, host pass over the copy constraints, 15 commitments, "
                        "16 coset transforms, Lagrange-basis key (group FFT) — host gate columns in pageable memory",
            "prove_ms_from_witness_table_pinned": round(t_prove * 1e3, 3),
            "witness_table_mb": round(len(cc["values"]) / 2**20, 1),
            "proof_blake2b": hashlib.blake2b(proof).hexdigest()[:32],
            "vk_blake2b": hashlib.blake2b(vk).hexdigest()[:32]}


def leaf_costs(ctx, log_n):
    """Seam-level drop-in cost, PCIe included (host buffers in, host buffers out): plonk_ntt / plonk_ntt_batch at the
    8n quotient-domain size and plonk_msm_batch of 4 scalar sets — what a Rust shim at the two crate-private
    seams (domain.rs:173-232, key.rs:384) pays per call when its vectors live in pinned memory
    (plonk_host_alloc / hipHostRegister).  The C entry points are called directly on the pinned addresses."""
    import ctypes

    import numpy as np
    rng = np.random.default_rng(3)
    n = 1 << log_n
    lib, h = ctx.lib, ctx.handle

    def rand_fr(cnt):
        a = rng.integers(0, 256, size=(cnt, 32), dtype=np.uint8)
        a[:, 31] &= 0x3F
        return a.tobytes()
    out = {}
    L8 = min(log_n + 3, 23)
    n8 = 1 << L8
    bufs = [plonk_amd.PinnedBuffer(32 * n8) for _ in range(5)]
    d = rand_fr(n8)
    for b in bufs:
        b.write(d)
    vp = ctypes.c_void_p
    ctx._check(lib.plonk_ntt(h, vp(bufs[0].ptr), L8, 0, 1, n8))           # warm-up (tables, staging)
    t0 = time.perf_counter()
    ctx._check(lib.plonk_ntt(h, vp(bufs[0].ptr), L8, 0, 1, n8))
    out["plonk_ntt_2p%d_ms" % L8] = round((time.perf_counter() - t0) * 1e3, 2)
    arr = (vp * 5)(*[vp(b.ptr) for b in bufs])
    ctx._check(lib.plonk_ntt_batch(h, arr, 5, L8, 0, 1, None))
    t0 = time.perf_counter()
    ctx._check(lib.plonk_ntt_batch(h, arr, 5, L8, 0, 1, None))
    out["plonk_ntt_batch5_2p%d_ms" % L8] = round((time.perf_counter() - t0) * 1e3, 2)
    # the way compute_coset_evaluations calls it (quotient_poly.rs:139-157): n + 3 coefficients in, the whole coset out
    lens = (ctypes.c_uint64 * 5)(*[min(n + 3, n8)] * 5)
    ctx._check(lib.plonk_ntt_batch(h, arr, 5, L8, 0, 1, lens))
    t0 = time.perf_counter()
    ctx._check(lib.plonk_ntt_batch(h, arr, 5, L8, 0, 1, lens))
    out["plonk_ntt_batch5_2p%d_coset_evaluations_ms" % L8] = round((time.perf_counter() - t0) * 1e3, 2)
    for b in bufs:
        b.free()
    sets = [plonk_amd.PinnedBuffer(32 * (n + 2)) for _ in range(4)]
    for sb in sets:
        sb.write(rand_fr(n + 2))
    sarr = (vp * 4)(*[vp(sb.ptr) for sb in sets])
    ms = (ctypes.c_uint64 * 4)(*[n + 2] * 4)
    res = ctypes.create_string_buffer(4 * 97)
    ctx._check(lib.plonk_msm_batch(h, sarr, ms, 4, res))
    t0 = time.perf_counter()
    ctx._check(lib.plonk_msm_batch(h, sarr, ms, 4, res))
    out["plonk_msm_batch4_2p%d_ms" % log_n] = round((time.perf_counter() - t0) * 1e3, 2)
    for sb in sets:
        sb.free()
    out["note"] = "pinned host buffers in and out, C entry points called directly (PCIe both ways included)"
    return out


VALU_PEAK_GWIPS = 1024 * 2.4 / 4          # G wave-instructions/s: 1024 SIMDs x 2.4 GHz, one VALU wave-instruction per 4 cycles


def valu_issue(wave_instructions, ms):
    """the bound that holds for every kernel on this path: integer-VALU issue"""
    ach = wave_instructions / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"achieved": round(ach, 1), "peak": VALU_PEAK_GWIPS, "unit": "G wave-instructions/s", "frac": round(ach / VALU_PEAK_GWIPS, 4)}


def accumulate_roofline(plan, model_wi, counter_wi, acc_ms_per_prove, pmc, run):
    """`roofline` of the dominant kernel.  counter_wi: VALU wave-instructions per proof from the committed SQ_INSTS_VALU pass
    (None when no round-6 profile is present or the workload is not the profiled one: the model count is used and says so);
    model_wi: non-zero digits / 64 lanes x 4850 instructions.  No field of this object can exceed 1: `frac` divides by the issue
    rate at the 2.4 GHz peak clock — above what the power-limited kernel ever gets — and `frac_at_measured_clock` is quoted from
    ONE counter pass (instructions, duration and clock of the same dispatches), not rescaled across runs as in round 5."""
    wi = counter_wi if counter_wi else model_wi
    issue = valu_issue(wi, acc_ms_per_prove)
    model = valu_issue(model_wi, acc_ms_per_prove)
    hbm_frac = round(run["achieved"] / HBM_PEAK_GBS, 5)
    out = {"bound": "valu-int-issue",
           # the variant that ran: 2^19 buckets (namespace nbl, lanes in order of length) above 2^18 terms over bit-position /
           # half-density rows, else 2^15 buckets (ordered lanes where slices are 32 entries long)
           "kernel": plan["accumulate_kernel"], "bucket_bits": plan["bucket_bits"], "digit_width": plan["digit_width"],
           "slice_entries": plan["slice_entries"],
           "achieved": issue["achieved"], "peak": issue["peak"], "unit": issue["unit"], "frac": issue["frac"],
           "frac_source": ("SQ_INSTS_VALU of the four prove() launches (%s) / this run's hipEvent time of the same launches" % pmc["src"]) if counter_wi
                          else "MODEL (no counter pass with valu_wave_instructions_per_proof for this workload): non-zero digits x 4850 instructions",
           "valu_wave_instructions_per_proof": None if not counter_wi else round(counter_wi),
           "valu_wave_instructions_per_launch": None if not pmc.get("instr_launch") else [round(v) for v in pmc["instr_launch"]],
           "model": {"wave_instructions_per_proof": round(model_wi), "instructions_per_addition_model": 4850,
                     "additions_per_scalar": run["digits_per_scalar"], "frac": model["frac"],
                     "ratio_to_counters": None if not counter_wi else round(model_wi / counter_wi, 4)},
           "traffic": pmc["traffic"], "valu_int_fraction": None if pmc["valu_busy"] is None else round(pmc["valu_busy"], 3),
           "hbm": {"achieved": round(run["achieved"], 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac},
           "hbm_frac": hbm_frac, "valu_issue_fraction": issue["frac"],
           "additions_per_scalar": run["digits_per_scalar"], "table_rows": run["table_rows"],
           "traffic_source": pmc["src"], "avg_launch_ms": round(run["avg_acc"], 4), "launches": int(run["acc_n"]),
           "algorithmic_bytes_per_launch": run["alg_bytes_per_launch"], "launch_groups_per_prove": list(run["groups"]),
           # the shader clock the counters show under this kernel (power-limited) and the issue fraction at THAT clock — both from the
           # committed counter pass, self-consistent
           "shader_clock_ghz_under_kernel": None if pmc["clock"] is None else round(pmc["clock"], 2),
           "frac_at_measured_clock": None if pmc.get("frac_clock") is None else round(pmc["frac_clock"], 4),
           "frac_vs_guide_valu_rate": round(issue["frac"] / 2, 4),
           "note": "integer-VALU bound (384-bit Montgomery products), not HBM bound.  `peak` = 1024 SIMDs x 2.4 GHz / 4 cycles: the "
                   "MEASURED issue class of v_mad_u64_u32 and the carry adds around it (4.4-4.9 cycles per wave-instruction, "
                   "profiles/r01/valu_issue_rates_gfx950.txt) — not the guide's generic VALU rate of one wave-instruction per 2 cycles, "
                   "against which the same kernel is at `frac_vs_guide_valu_rate`.  `frac_at_measured_clock`: the same counter pass's instructions "
                   "over ITS duration at ITS shader clock (power-limited, well below 2.4 GHz).  `traffic` (PMC, HBM bytes per launch) is one "
                   "128-B table gather per non-zero digit by design; see DESIGN.md 4.2 / 6"}
    # no fraction above 1 ships (round 5 printed frac_at_measured_clock = 1.02): a value outside [0, 1] means the counter
    # pass and this run do not describe the same kernel — it is withheld and named, never printed as if it were a measurement
    for k in ("frac", "hbm_frac", "valu_issue_fraction", "frac_at_measured_clock", "frac_vs_guide_valu_rate", "valu_int_fraction"):
        if out[k] is not None and not 0.0 <= out[k] <= 1.0:
            out.setdefault("withheld_not_a_fraction", {})[k] = out[k]
            out[k] = None
    return out


def quotient_roofline(prover, n, qmul, has_pi, profile, ms_per_launch):
    nz = getattr(build_prover, "selectors_nonzero", 5)
    """Algorithmic bytes of the point-wise quotient pass: every array the kernel reads once per point + the one it writes,
    32 B each (poly.hip quotient_kernel): wires a b c d, z (read at i and at the rotated index: counted once), q_c, q_arith,
    `linear`, sigma 1-4, L1, the non-zero ones of q_m q_l q_r q_o q_f, the public-input evaluations when there are any, and —
    with custom gates — the rotated wires come from the same arrays; the widget selectors add one array each."""
    arrays = 5 + 2 + 1 + 4 + 1 + nz + (1 if has_pi else 0) + (4 if profile == "widgets" else 0)
    bytes_per_point = 32 * (arrays + 1)
    pts = qmul * n
    ach = bytes_per_point * pts / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
    # what binds it (profiles/r03e section 6): ~7 000 VALU instructions per point without custom gates (28 Fr29 products + 20
    # limb conversions), ~32 000 with every widget — issue-bound at a third of the HBM peak
    instr = 32000 if profile == "widgets" else 7000
    issue = valu_issue(pts * instr / 64, ms_per_launch)
    return {"bound": "valu-int-issue", "kernel": "quotient_kernel", "points": pts, "instructions_per_point": instr,
            "achieved": issue["achieved"], "peak": issue["peak"], "unit": issue["unit"], "frac": issue["frac"],
            "algorithmic_bytes_per_point": bytes_per_point, "arrays_read": arrays, "avg_launch_ms": round(ms_per_launch, 4),
            "hbm": {"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)},
            "hbm_frac": round(ach / HBM_PEAK_GBS, 4),
            "note": "hipEvent pair around the launch inside the timed region (slot 3)"}


def ntt_roofline(ctx, log_n, qd8):
    """The transforms of one prove() timed on their own (device-resident, same entry point the prover uses): the
    quotient-domain coset NTT from n + 3 coefficients, the inverse coset NTT and the size-n inverse NTT.  Algorithmic
    bytes 64 N per transform whatever the number of passes."""
    n = 1 << log_n
    Lq = log_n + (3 if qd8 else 2)
    out = {}
    for name, L, inv, coset, in_len in (("coset_ntt_quotient_domain", Lq, False, True, n + 3),
                                        ("coset_intt_quotient_domain", Lq, True, True, 1 << Lq),
                                        ("intt_n", log_n, True, False, n)):
        N = 1 << L
        src, dst, tmp = ctx.alloc(32 * N), ctx.alloc(32 * N), ctx.alloc(32 * N)
        src.upload(bytes(32 * min(N, 1 << 16)))      # contents do not matter for the timing (no data-dependent branches)
        ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, in_len)
        ctx.sync()
        iters = 10
        t0 = time.perf_counter()
        for _ in range(iters):
            ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, in_len)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / iters
        for b in (src, dst, tmp):
            b.free()
        # N/2 log2 N butterflies of ~290 VALU instructions (fr29.cuh) + one inter-pass twiddle product per
        # element and pass boundary (~150): the issue-rate figure that bounds the passes
        npass = 1 if L <= 10 else (2 if L <= 18 else 3)
        issue = valu_issue((N // 2 * L * 290 + N * (npass - 1) * 150) / 64, ms)
        out[name] = {"log_size": L, "ms": round(ms, 4), "melem_per_s": round(N / ms / 1e3, 1),
                     "achieved": issue["achieved"], "peak": issue["peak"], "unit": issue["unit"], "frac": issue["frac"],
                     "hbm": {"achieved": round(64 * N / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(64 * N / ms / 1e6 / HBM_PEAK_GBS, 4)},
                     "hbm_frac": round(64 * N / ms / 1e6 / HBM_PEAK_GBS, 4)}
    elog = ctx.get_config().ntt_elements_log2 or 2   # ntt.hip: 4 elements per lane by default (1024-element tiles, four waves per SIMD)
    return {"bound": "valu-int-issue", "kernel": f"ntt_pass_kernel<R, T, ELOG = {elog}> (2-3 passes per transform)",
            "elements_per_lane": 1 << elog, "algorithmic_bytes": "64 N per transform",
            "note": "timed standalone with wall clock around 10 back-to-back launches; integer-VALU bound (Fr29 butterflies), "
                    "a k-pass plan moves k x 64 N actual bytes; inside prove() the side-stream transforms issued under a busy "
                    "MSM pipeline use the 8-elements-per-lane kernels (DESIGN.md 4.1)", "transforms": out}


def micro_scalars(m: int, profile: str, seed: int) -> bytes:
    """m scalars as Montgomery limb bytes: `uniform` (below 2^254, i.e. uniform field elements for every purpose of an MSM) or
    `bench-like` (half of the values < 4 — Montgomery forms of 0 .. 3 — half uniform)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    a[:, 31] &= 0x3F
    if profile == "bench-like":
        table = np.array([list(int.to_bytes(v * (1 << 256) % Q, 32, "little")) for v in range(4)], dtype=np.uint8)
        small = rng.random(m) < 0.5
        a[small] = table[rng.integers(0, 4, size=m)[small]]
    return a.tobytes()


def msm_micro_rows(ctx, log_m: int, reps: int = 3):
    """SURVEY 8(d) "MSM micro": m = 2^log_m + 6 scalars against the commit key the context holds, batches of 1 and 4 over the
    same bases, `uniform` and `bench-like` scalars.  Called through plonk_msm_batch on pinned host scalars; `device_ms` is the
    time of the MSM kernels between hipEvents on the library's stream (slots 1 + 2: sort, accumulation, reduction — no PCIe),
    `wall_ms` the whole call (scalar upload and the host's finish of the bit sums included)."""
    import ctypes
    m = (1 << log_m) + 6
    lib, h, vp = ctx.lib, ctx.handle, ctypes.c_void_p
    rows = []
    for prof in ("uniform", "bench-like"):
        bufs = [plonk_amd.PinnedBuffer(32 * m) for _ in range(4)]
        for k, b in enumerate(bufs):
            b.write(micro_scalars(m, prof, 0x5EED0100 + 16 * log_m + k))
        arr = (vp * 4)(*[vp(b.ptr) for b in bufs])
        ms = (ctypes.c_uint64 * 4)(*[m] * 4)
        res = ctypes.create_string_buffer(4 * 97)
        for batch in (1, 4):
            ctx._check(lib.plonk_msm_batch(h, arr, ms, batch, res))      # warm-up: work buffers, staging
            ctx.profile(True)
            ctx.profile_reset()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx._check(lib.plonk_msm_batch(h, arr, ms, batch, res))
            wall = (time.perf_counter() - t0) * 1e3 / reps
            acc, _ = ctx.profile_read(1)
            oth, _ = ctx.profile_read(2)
            ctx.profile(False)
            plan = ctx.last_msm()
            dev = (acc + oth) / reps
            rows.append({"m": m, "batch": batch, "scalars": prof, "device_ms": round(dev, 3), "accumulate_ms": round(acc / reps, 3),
                         "wall_ms_pinned_host_scalars": round(wall, 3), "mscalar_per_s": round(batch * m / max(dev, 1e-9) / 1e3, 1),
                         "algorithmic_gb_per_s": round((32 * batch + 96) * m / max(dev, 1e-9) / 1e6, 1),
                         "kernel": plan["accumulate_kernel"], "table_rows": plan["table_rows"], "bucket_bits": plan["bucket_bits"],
                         "digit_width": plan["digit_width"]})
        for b in bufs:
            b.free()
    return rows


def ntt_micro(ctx):
    """SURVEY 8(d) "NTT micro": N = 2^12 .. 2^25, forward / inverse / coset-forward, device-resident, wall clock around back-to-back
    launches (timing does not depend on the data).  Algorithmic bytes 64 N per transform."""
    rows = []
    for L in (12, 16, 20, 22, 23, 25):
        N = 1 << L
        src, dst, tmp = ctx.alloc(32 * N), ctx.alloc(32 * N), ctx.alloc(32 * N)
        src.upload(bytes(32 * min(N, 1 << 16)))
        for name, inv, coset in (("forward", False, False), ("inverse", True, False), ("coset_forward", False, True)):
            ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, N)
            ctx.sync()
            iters = 20 if L <= 20 else (8 if L <= 23 else 4)
            t0 = time.perf_counter()
            for _ in range(iters):
                ctx.ntt_dev(src.ptr, dst.ptr, tmp.ptr, L, inv, coset, N)
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3 / iters
            npass = 1 if L <= 10 else (2 if L <= 18 else 3)
            issue = valu_issue((N // 2 * L * 290 + N * (npass - 1) * 150) / 64, ms)
            rows.append({"log_size": L, "transform": name, "ms": round(ms, 4), "melem_per_s": round(N / ms / 1e3, 1),
                         "hbm_gb_per_s": round(64 * N / ms / 1e6, 1), "hbm_frac": round(64 * N / ms / 1e6 / HBM_PEAK_GBS, 4),
                         "valu_issue_frac": issue["frac"], "passes": npass})
        for b in (src, dst, tmp):
            b.free()
    return rows


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_small(ctx, log_n: int, blinders):
    """The same port at another size north_star names (2^16, 2^12): one whole prove() on the host cores, and the SAME circuit,
    key and blinders proved on the GPU for the byte comparison."""
    dg = {}
    prover, wbuf, _ = build_prover(ctx, log_n, 0, 1, None, "dense")
    gpu_proof = prover.prove_dev(wbuf.ptr, prover.public_inputs, blinders)
    vk48 = prover.vk_commitments()
    prover.close()
    wbuf.free()
    rec, cpu_proof = cpu_baseline(log_n, vk48)
    rec["proof_matches_gpu"] = bool(cpu_proof == gpu_proof)
    return rec


def cpu_baseline(log_n: int, vk48: bytes):
    """One WHOLE prove() on the host CPU with the C restatement of the reference's algorithm
    (oracle/c/oracle_prove.c: best_fft, arkworks-style Pippenger, 8n quotient, OpenMP in place of rayon),
    every core the container may use.  Sizes above 2^20 are measured at 2^20 and scaled linearly."""
    from oracle import cbind
    sample_log = min(log_n, 20)
    n = 1 << sample_log
    threads = cbind.max_threads()
    wires, cols, trivial = BC.arithmetic_circuit(sample_log, "dense")
    mont = plonk_amd.fr_to_bytes_mont
    polys = {k: mont(v) for k, v in trivial.items()}
    for name, raw in cols.items():
        polys[name] = cbind.ntt_bytes(raw, sample_log, True, False, n, threads)
    srs = cbind.srs_generate(mont([TAU]), mont([G_SCALAR]), n + 7, threads)
    cp = cbind.CProver(n, b"bench", polys, srs, vk48=vk48 if sample_log == log_n else None, threads=threads)
    bl = mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
    t0 = time.perf_counter()
    proof = cp.prove(wires, [], b"", bl)
    dt = time.perf_counter() - t0
    sec = dict(cp.seconds)
    cp.close()
    scale = float(1 << (log_n - sample_log))
    return {"value": round(dt * 1e3 * scale, 1), "unit": "ms", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "proof_blake2b": hashlib.blake2b(proof).hexdigest()[:32],
            "sample": (f"one whole prove() of the bench circuit at 2^{sample_log} gates with the C restatement of the "
                       f"reference algorithm (oracle/c, OpenMP, {threads} threads): {dt * 1e3:.0f} ms = NTT {sec['ntt'] * 1e3:.0f} + "
                       f"MSM {sec['msm'] * 1e3:.0f} + quotient {sec['quotient'] * 1e3:.0f} + grand product {sec['perm'] * 1e3:.0f} + "
                       f"round 4-5 {sec['tail'] * 1e3:.0f} ms" + (f", scaled x{scale:g}" if scale != 1 else "") +
                       "; not the Rust binary (no cargo in this image)")}, proof


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, rank r on device r) with
    the environment torch.distributed.run would give them, forward rank 0's stdout (the JSON line), and supervise:
    when one rank fails the others — which may be waiting for it inside a collective — are terminated."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench launcher] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def dry_launch() -> int:
    """every rank: join the control-plane group, check that all N ranks are there; rank 0 prints one JSON line"""
    import datetime

    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if os.environ.get("PLONK_BENCH_DRY_FAIL_RANK") == str(rank):   # test hook: this rank dies before it joins
        return 3
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    ok = int(t.item()) == world * (world + 1) // 2
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": world if ok else -1,
                          "local_rank": int(os.environ.get("LOCAL_RANK", "-1"))}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-gates", type=int, default=20)
    ap.add_argument("--profile", default="dense", choices=["dense", "bench-like", "widgets"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-2p22", action="store_true", help="skip the 2^22-gate extra of the default line (about half a minute of setup)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads / leaf costs of the N=1 line")
    ap.add_argument("--host-wires-only", action="store_true", help="of the extras, only the proof from pinned host wire columns (A/B scripts)")
    ap.add_argument("--from-circuit", action="store_true",
                    help="build the prover with plonk_compile from gate columns (Compiler::preprocess on the device) instead of "
                         "from coefficient forms, and check plonk_prover_prove_witnesses against the column entry point")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher self-test: every rank joins the gloo group, all-reduces its rank and exits (no GPU needed)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))       # `python bench.py --gpus N`: this process only launches and supervises the N ranks
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:                 # torchrun --nproc-per-node N ... bench.py --gpus M: refuse to print a mislabelled line
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch N ranks for --gpus N "
                 "(python bench.py --gpus N starts them itself)")
    if args.dry_launch:
        sys.exit(dry_launch())

    # stdout carries exactly ONE line (the JSON): everything else this process or its libraries print — RCCL's version
    # banner sits in the C stdio buffer until exit and would land AFTER the JSON line — goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    log_n = args.log_gates
    gpu_config = plonk_amd.GpuConfig()   # all defaults (include/plonk_hip.h plonk_gpu_config); the line reports the effective values
    if os.environ.get("PLONK_BENCH_SHARE_GPU") == "1":   # several ranks on ONE GPU (functional test on a 1-GPU box)
        local_rank = 0
        # Every rank sizes its tables for a device of its own ("at most half of the free HBM"): W ranks on one device would
        # each take a row per bit position for both of their keys — 8 x 32 GiB at 2^22 gates / W = 8 — and run it out of memory.
        # Sharing ranks therefore agree on the densest layout whose W copies fit in ~half of the device: every second bit
        # position (the same 2^19-bucket kernels), else window rows.  A real multi-GPU run never takes this branch.
        if world > 1 and "PLONK_MSM_TABLE" not in os.environ:
            per_rank_points = ((1 << log_n) + 7 + world - 1) // world
            bitpos_bytes = world * 2 * 256 * 128 * per_rank_points          # commit key + Lagrange-basis slice, 256 rows of 128 B
            if per_rank_points > (1 << 18) + 64 and bitpos_bytes > 120 << 30:
                gpu_config.table_mode = plonk_amd.TABLE_HALFPOS if bitpos_bytes // 2 <= 140 << 30 else plonk_amd.TABLE_WINDOW
    ctx = plonk_amd.Context(local_rank, gpu_config)
    dist = None
    allgather = None
    collective = None
    n_ranks_rccl = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        import datetime
        dist.init_process_group(backend="gloo", rank=rank, world_size=world,     # control plane only
                                timeout=datetime.timedelta(seconds=int(os.environ.get("PLONK_BENCH_JOIN_TIMEOUT_S", "600"))))
        want_rccl = os.environ.get("PLONK_BENCH_BACKEND", "nccl") == "nccl"
        ok = 0
        if want_rccl:
            # PLONK_BENCH_TRANSPORT_LIBRARY (set by the tests only): another file with the nccl* entry points — the stand-in of
            # tests/fake_rccl, which lets ranks that SHARE one GPU run the library's device-pointer collectives
            tl = os.environ.get("PLONK_BENCH_TRANSPORT_LIBRARY")
            if tl:
                plonk_amd.Context.comm_set_library(tl)
            # rank 0 creates the ncclUniqueId, gloo broadcasts it, every rank joins with its context
            box = [None]
            if rank == 0:
                try:
                    box[0] = plonk_amd.Context.comm_unique_id()
                except Exception as e:   # noqa: BLE001
                    print(f"[bench] ncclGetUniqueId failed: {e}", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                try:
                    ctx.comm_init(box[0], rank, world)
                    ctx.comm_selftest()
                    ok = 1
                except Exception as e:   # noqa: BLE001
                    print(f"[bench rank {rank}] RCCL bring-up failed ({type(e).__name__}: {e})", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # unanimous, so that no rank waits in the other transport
            ok = int(flag.item())
            if not ok:
                try:
                    ctx.comm_destroy()
                except Exception:   # noqa: BLE001
                    pass
        collective = "gloo"
        if ok:
            # what the library actually loaded decides the label: only a librccl file may be called "rccl", and only its
            # communicator's size is reported as n_ranks_rccl — a stand-in can never pass for hardware collectives
            loaded = os.path.basename(plonk_amd.Context.comm_library())
            if loaded.startswith("librccl"):
                collective, n_ranks_rccl = "rccl", ctx.comm_info()[1]
            else:
                collective = "stand-in:" + loaded
                assert ctx.comm_info() == (rank, world)
        if not ok:
            def allgather(send: bytes) -> bytes:   # host-callback transport (tests / fallback)
                t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
                out = torch.empty(world * t.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, t)
                return out.numpy().tobytes()

    t_setup = time.perf_counter()
    prover, wbuf, srs_total = build_prover(ctx, log_n, rank, world, allgather, args.profile, args.from_circuit)
    t_setup = time.perf_counter() - t_setup
    witness_values = build_prover.witness_values
    pi = prover.public_inputs
    blinders = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    proof = None
    for _ in range(args.warmup):
        proof = prover.prove_dev(wbuf.ptr, pi, blinders)
    if witness_values is not None:   # every rank: the proof from the witness table equals the proof from the wire columns
        assert prover.prove_witnesses(witness_values, pi, blinders) == prover.prove_dev(wbuf.ptr, pi, blinders)
        if proof is None:
            proof = prover.prove_dev(wbuf.ptr, pi, blinders)
    # timed region: exactly K proofs, hipEvent pairs recorded around the dominant kernels
    ctx.profile(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = prover.prove_dev(wbuf.ptr, pi, blinders)
    barrier()
    elapsed = time.perf_counter() - t0
    acc_ms, acc_n = ctx.profile_read(1)     # msm_accumulate launches inside the timed region
    oth_ms, oth_n = ctx.profile_read(2)
    q_ms, q_n = ctx.profile_read(3)
    rep_ms, rep_n = ctx.profile_read(4)
    ctx.profile(False)
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ph = torch.frombuffer(bytearray(proof), dtype=torch.uint8).to(torch.int32)   # all ranks must hold the identical proof
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
        # HBM traffic / VALU utilisation of the same kernel come from PMC passes (rocprofv3 --pmc, separate runs):
        # NOT measured by this run — the latest committed profile is quoted with its source
        valu_busy = traffic = pmc_src = pmc_clock = pmc_instr = pmc_instr_launch = pmc_frac_clock = None
        if world == 1 and log_n == 20 and args.profile == "dense":
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc.json"))):
                try:
                    pj = json.load(open(f))
                    traffic = round(pj["traffic_bytes_per_launch"])
                    valu_busy = pj.get("valu_busy_frac")
                    pmc_clock = pj.get("shader_clock_ghz_under_kernel")
                    # round 6: SQ_INSTS_VALU of the four prove() launches of the dominant kernel (tools/summarize_profile.py) — what
                    # `achieved` / `frac` are built on; absent in the profiles of rounds 1-5
                    pmc_instr = pj.get("valu_wave_instructions_per_proof")
                    pmc_instr_launch = pj.get("valu_wave_instructions_per_launch")
                    pmc_frac_clock = pj.get("valu_issue_frac_at_measured_clock")
                    pmc_src = os.path.relpath(f, ROOT) + " (rocprofv3 --pmc passes of an earlier run, not this run)"
                except Exception:   # noqa: BLE001
                    pass
        eff = ctx.get_config()              # the EFFECTIVE configuration (struct + A/B overrides), as the library resolved it
        qd8 = eff.quotient_domain == 8 or world == 8
        table_rows = ctx.table_rows()
        # what the library says a commitment group of this size runs as over the commit key (the same function its launcher
        # follows: plonk_ctx_describe_msm) — kernel variant, bucket count, digit width
        plan = ctx.describe_msm(m_local, 4)
        w_bits, nb = plan["digit_width"], 1 << plan["bucket_bits"]
        # additions per scalar, uniform scalars: 16 signed 16-bit windows; width-w NAF digits over bit-position rows: 254.9 / (w + 1)
        # + 1/2; even-position digits over half-density rows: 254.9 / (w + 2/3) + 1/2 — minus the first entry of every bucket's
        # slice where a bucket is one slice (an assignment, not an addition: 2^19 buckets)
        if table_rows == 16:
            digits_per_scalar = 16.0 - 0.5
        else:
            digits = 254.86 / (w_bits + (1 if table_rows == 256 else 2 / 3)) + 0.5   # (round 5 called this `per` too and garbled config.srs)
            digits_per_scalar = round(digits - (nb / max(m_local, 1) if plan["bucket_bits"] > 15 else 0.5), 2)
        npoly = 6 if pi else 5
        out = {
            "metric": "prove() wall-clock (ms) at 2^%d gates" % log_n,
            "value": round(ms_per_step, 3), "unit": "ms", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32 limbs (Fr 256-bit / Fp 384-bit Montgomery)",
            "data": "synthetic",
            # the contract's `value`: inputs resident in HBM when the timed region starts.  What a drop-in caller of Prover::prove
            # pays on top is printed beside it: prove_ms_host_wires_pinned (2^20) / host_legs_2p22
            "value_excludes": "H2D of the 4 x 32 n wire bytes (%d MiB at this size)" % (4 * 32 * n >> 20),
            "config": {"workload": "Prover::prove V3, synthetic 2^%d-gate circuit, profile `%s` (%s), random SRS of n+7 points, "
                                   "wires/ProverKey/SRS tables resident in HBM" % (
                                       log_n, args.profile,
                                       {"dense": "arithmetic gates, uniformly random selectors and wires, no public inputs",
                                        "bench-like": "arithmetic gates, half of the wire values < 4, no public inputs",
                                        "widgets": "range + logic + fixed-base + curve-addition + arithmetic gates, 2 public inputs"}[args.profile]),
                       "gates": n,
                       "ntt": ("%d iNTT(n) + %d cosetNTT(%dn) + 1 cosetiNTT(%dn)" % (npoly, npoly, 8 if qd8 else 4, 8 if qd8 else 4)) +
                              ("" if qd8 else " [quotient interpolated on the 4n coset + de-aliasing; reference: 8n]") +
                              (" — per rank: the coset transforms as size-n transforms on its residue classes" if world in (2, 4, 8) else ""),
                       "msm": "11 x ~n terms",
                       "parallelism": ("1 GPU" if world == 1 else
                                       ("x%d: MSM by SRS point range%s; quotient by coset residue class; rounds 4-5 by coefficient range" % (
                                           world, " (wire commitments by whole column)" if getattr(build_prover, "wire_split", "") == "commitment" else ""))
                                       if world in (2, 4, 8) else "x%d: MSM by SRS point range" % world),
                       "srs": "rank's point range streamed from pinned host memory in 2^18-point chunks (upload of chunk k+1 under "
                              "the window-table build of chunk k): %d points in %.2f s" % (min(per, srs_total), build_prover.srs_stream_s),
                       "collective": collective, "n_ranks_rccl": n_ranks_rccl, "setup_s": round(t_setup, 1),
                       "gpu_config": eff.as_dict(),   # plonk_gpu_config as the library resolved it (defaults + A/B overrides)
                       "prover_built_by": "plonk_compile (gate columns)" if args.from_circuit else "plonk_prover_create (coefficient forms)"},
            # whole-job MSM rate: all 11 x (n + 6) terms of a proof over the time rank 0 spends in its (sharded) MSM kernels
            "msm_mscalar_per_s": round(11 * (n + 6) / max((acc_ms + oth_ms) / args.steps, 1e-9) / 1e3, 2),
            "proof_blake2b": hashlib.blake2b(proof).hexdigest()[:32],
            # the dominant kernel is bound by integer-VALU issue, not by HBM: `frac` is the issue-rate fraction.
            # Round 6: `achieved` = the VALU wave-instructions the COUNTERS show for the four prove() launches
            # (SQ_INSTS_VALU, profiles/<tag>/pmc.json `valu_wave_instructions_per_proof`, same build) over this run's hipEvent time of
            # those launches; `peak` = one VALU wave-instruction per 4 cycles per SIMD.  The static count of rounds 2-5 (non-zero digits x
            # 4850 instructions per mixed addition, profiles/r02c/accumulate_isa.md) stays as `model` with its ratio to the counters.
            # The HBM figure the contract asks for (algorithmic (32 b + 96) m bytes per group launch over the launch time) is `hbm`.
            "roofline": accumulate_roofline(plan, digits_per_scalar * 11 * (n + 6) / max(world, 1) / 64 * 4850, pmc_instr, acc_ms_per_prove,
                                            dict(traffic=traffic, valu_busy=valu_busy, clock=pmc_clock, frac_clock=pmc_frac_clock, src=pmc_src,
                                                 instr_launch=pmc_instr_launch),
                                            dict(achieved=achieved, digits_per_scalar=digits_per_scalar, table_rows=table_rows, avg_acc=avg_acc,
                                                 acc_n=acc_n, alg_bytes_per_launch=alg_bytes_per_prove // len(groups), groups=groups)),
            # the pass of a proof that comes closest to HBM: quotient_kernel over the quotient-domain points
            "roofline_quotient": quotient_roofline(prover, n, 8 if qd8 else 4, bool(pi), args.profile, q_ms / max(q_n, 1)),
            "kernel_ms_per_prove": {"msm_accumulate": round(acc_ms / args.steps, 3),
                                    "msm_other": round(oth_ms / args.steps, 3),
                                    "quotient_pointwise": round(q_ms / args.steps, 3),
                                    # wire / permutation polynomials of rounds 1-2: the part every rank of a multi-GPU run repeats
                                    "rounds_1_2_polynomials_replicated": round(rep_ms / args.steps, 3)},
        }
        vk48 = prover.vk_commitments()
        if world == 1 and not args.no_extras:
            try:   # the boundary handing over HOST wire columns (pinned): PCIe-inclusive prove(), never `value`
                out["prove_ms_host_wires_pinned"] = host_wire_legs(ctx, prover, wbuf, n, blinders, proof, max(2, min(args.steps, 5)))
                # what separates it from `value`: the first column's copy (measured, nothing can run under it) + three extra
                # launch sets (columns a, b, then c + d: DESIGN.md 4.4)
                out["host_wires"] = {"first_column_h2d_ms": host_wire_legs.first_column_h2d_ms,
                                     "pcie_floor_ms": round(ms_per_step + host_wire_legs.first_column_h2d_ms, 3),
                                     "gap_to_value_ms": round(out["prove_ms_host_wires_pinned"] - ms_per_step, 3),
                                     "wire_bytes_mib": 4 * 32 * n >> 20}
            except Exception as e:   # noqa: BLE001
                out["host_wires_error"] = repr(e)
        if world == 1 and not args.no_extras and not args.host_wires_only and args.profile == "dense":
            try:   # seam-level cost, measured while the prover is still alive: once a process has FREED tens of GB of device
                   # buffers the runtime stops overlapping the two copy directions of plonk_ntt_batch (33 -> 53 ms; bisected in
                   # profiles/r03b/ntt_batch_overlap_bisect.txt — variant C = prover.close() before the call)
                out["roofline_ntt"] = ntt_roofline(ctx, log_n, qd8)
                out["leaf_ms"] = leaf_costs(ctx, log_n)
                out["ntt_micro"] = ntt_micro(ctx)
                if log_n >= 16:   # SURVEY 8(d) MSM micro rows at this size (the context still holds the timed run's commit key)
                    out["msm_micro"] = msm_micro_rows(ctx, log_n)
            except Exception as e:   # noqa: BLE001
                out["leaf_error"] = repr(e)
        prover.close()
        wbuf.free()
        if world == 1 and not args.no_extras and not args.host_wires_only and args.profile == "dense":
            try:   # the other workloads of SURVEY §8(d); never a reason to lose the line
                k = max(2, min(args.steps, 5))
                out["prove_ms_bench_like"] = time_profile(ctx, log_n, "bench-like", k, blinders)
                out["prove_ms_all_widgets_pi"] = time_profile(ctx, log_n, "widgets", k, blinders)
                # north_star: prove-time at 2^16 .. 2^22 (and BASELINE config 1's 2^12) on the driver's clock, each with its digest
                for lg, reps in ((12, 6 * k), (16, 4 * k)):
                    if log_n != lg:
                        dg = {}
                        out["prove_ms_2p%d" % lg] = time_profile(ctx, lg, "dense", reps, blinders, dg)
                        out["proof_blake2b_2p%d" % lg] = dg.get("proof_blake2b")
                if log_n != 16:   # MSM micro at 2^16 + 6 terms: a key of that size (window rows, 2^15 buckets)
                    pts = ctx.alloc(96 * ((1 << 16) + 7))
                    ctx.srs_generate_dev(TAU, G_SCALAR, (1 << 16) + 7, pts.ptr)
                    ctx.srs_load_dev(pts.ptr, (1 << 16) + 7)
                    pts.free()
                    out["msm_micro"] = msm_micro_rows(ctx, 16, 10) + out.get("msm_micro", [])
                out["compile"] = compile_costs(ctx, log_n if log_n <= 20 else 20, blinders)
                if log_n <= 20:   # same circuit, key and blinders as the timed run, built the other way
                    out["compile"]["proof_matches_timed_run"] = bool(out["compile"]["proof_blake2b"] == out["proof_blake2b"])
            except Exception as e:   # noqa: BLE001
                out["extras_error"] = repr(e)
        if world == 1 and not args.no_extras and not args.host_wires_only and args.profile == "dense" and log_n == 20 and not args.no_2p22:
            try:   # BASELINE config 5's size on one GPU (commit key streamed from pinned host memory): after everything else, own guard
                t22 = time.perf_counter()
                dg = {}
                legs = {}
                try:     # through plonk_compile: the one prover that can start a proof from resident columns, host columns AND the witness table
                    out["prove_ms_2p22"] = time_profile(ctx, 22, "dense", 3, blinders, dg, host_legs=legs, from_circuit=True)
                except Exception as e:   # noqa: BLE001   (the coefficient-form constructor of rounds 3-5)
                    legs = {"compile_path_error": repr(e)}
                    out["prove_ms_2p22"] = time_profile(ctx, 22, "dense", 3, blinders, dg, host_legs=legs)
                out["prove_2p22_prover_built_by"] = dg.get("prover_built_by")
                out["proof_blake2b_2p22"] = dg.get("proof_blake2b")
                out["host_legs_2p22"] = legs   # the drop-in costs at BASELINE config 5's size
                out["msm_micro"] = out.get("msm_micro", []) + msm_micro_rows(ctx, 22, 2)   # the 2^22 + 7-point key is still loaded
                out["prove_2p22_setup_and_run_s"] = round(time.perf_counter() - t22, 1)
            except Exception as e:   # noqa: BLE001
                out["prove_2p22_error"] = repr(e)
            try:   # give the 137 GB of bit-position rows back before the CPU leg
                ctx.srs_load([])
            except Exception:   # noqa: BLE001
                pass
        if world == 1 and not args.no_cpu_baseline and args.profile == "dense":
            try:
                out["cpu_baseline"], cpu_proof = cpu_baseline(log_n, vk48)
                if log_n <= 20:   # same SRS, circuit, witness and blinders: the CPU port must produce the same bytes
                    out["cpu_baseline"]["proof_matches_gpu"] = bool(cpu_proof == proof)
                if not args.no_extras:   # north_star: "prove-time at 2^16 .. 2^22 ... next to the CPU path": the port at the small sizes too
                    for lg in (16, 12):
                        if lg != log_n:
                            out["cpu_baseline_2p%d" % lg] = cpu_baseline_small(ctx, lg, blinders)
            except Exception as e:  # the baseline is a report, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "ms", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
