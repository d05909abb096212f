"""GPU, several ranks on ONE device: the multi-GPU prove() of prover.hip (prover_prove_sharded) must output
the same Proof bytes as the single-GPU run — MSMs sharded by SRS point range, the quotient by residue
class of the coset, rounds 4-5 by coefficient range.  The exchanges go through the library's host-callback
transport over gloo here (RCCL refuses two ranks on one device): the fallback a context without a communicator takes.
The transport a multi-GPU node takes — nccl* collectives on device pointers inside the library — runs with real peers in
tests/test_gpu_standin_transport.py (round 5; most of this file's matrix moved there), and with the real librccl and a
one-rank communicator below (dlopen, communicator, both collectives on the library's stream)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


_single = {}


def single(log_gates, profile):
    key = (log_gates, profile)
    from session_cache import SINGLE
    if key not in _single and key in SINGLE:     # tests/test_gpu_fullsize.py proved (and checked) exactly this circuit earlier in the session
        _single[key] = SINGLE[key]
    if key not in _single:
        _single[key] = _run([sys.executable, "bench.py", "--log-gates", str(log_gates), "--steps", "1", "--warmup", "0",
                             "--profile", profile, "--no-cpu-baseline", "--no-extras"])
    return _single[key]


def multi(ranks, log_gates, profile, env, extra=()):
    return _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
                 "--master-addr", "127.0.0.1", "--master-port", str(29600 + ranks), "bench.py", "--gpus", str(ranks),
                 "--log-gates", str(log_gates), "--steps", "1", "--warmup", "1", "--profile", profile, "--no-extras", *extra], env)


@pytest.mark.parametrize("ranks,log_gates,profile", [(2, 13, "widgets"), (8, 13, "widgets")])
def test_sharded_quotient_prove_matches_single_gpu(ranks, log_gates, profile):
    """world in {2, 8}: class-sharded quotient (Q = 4 classes, 8 for world 8), every widget + public inputs, through the
    HOST-CALLBACK transport (plonk_prover_desc.allgather over gloo).  Round 5: the other world sizes / sizes of this matrix —
    (4, 13), (4, 16), (2, 12) and the 2^20 / 2^22 cases — run through the library's own device collectives instead
    (tests/test_gpu_standin_transport.py), which is the path a multi-GPU node takes."""
    s = single(log_gates, profile)
    m = multi(ranks, log_gates, profile, {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1"})
    assert m["n_gpus"] == ranks and s["n_gpus"] == 1 and m["config"]["collective"] == "gloo"
    assert "residue class" in m["config"]["parallelism"]
    assert m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.slow
@pytest.mark.parametrize("ranks", [4])
def test_sharded_prove_matches_single_gpu_at_2p20(ranks):
    """BASELINE config 4's size: the sharded proof of the 2^20-gate bench circuit — W = 2 (Q = 4 classes; ranks of 2^19 points:
    2^19 buckets), W = 4 (2^18 points: window rows, ordered 32-entry slices; sharded grand product) and W = 8
    (Q = 8: the 8n class layout) — must be the single-GPU proof byte for byte; the single-GPU bytes at this size are
    compared with the C oracle in tests/test_gpu_prove_sizes.py.  Launched as `python bench.py --gpus N`, i.e. through
    bench.py's own rank launcher (the ranks share this box's one GPU)."""
    s = single(20, "dense")
    m = _run([sys.executable, "bench.py", "--gpus", str(ranks), "--log-gates", "20", "--steps", "1", "--warmup", "0", "--no-extras"],
             {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1"})
    assert m["n_gpus"] == ranks and m["config"]["collective"] == "gloo"
    assert m["proof_blake2b"] == s["proof_blake2b"]


def _run_failing(cmd, env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0, out.stdout[-2000:]
    return out.stdout + out.stderr


def test_ranks_that_disagree_on_the_wire_commitment_mode_are_refused():
    """rank 0 hands the prover the WHOLE Lagrange-basis key (wire group by column), rank 1 its slice (by point
    range).  Until round 6 each rank inferred its mode from its own lagrange_count and nothing compared them: the all-gather
    would have added whole-column commitments to point-range partial sums — wrong a / b / c / d that no identity check sees.
    Now plonk_prover_create all-gathers one mode byte per rank and every rank fails with PLONK_ERR_ARG."""
    log = _run_failing([sys.executable, "bench.py", "--gpus", "2", "--log-gates", "12", "--steps", "1", "--warmup", "0", "--no-extras"],
                       {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1", "PLONK_BENCH_WIRE_SPLIT": "commitment,range"})
    assert "disagree on the wire-commitment mode" in log, log[-3000:]


@pytest.mark.parametrize("split", ["range", "commitment"])
def test_a_rank_with_a_wrong_lagrange_key_is_caught_at_prover_creation(split):
    """Round 6: the Lagrange-key identity (an MSM of random values over the supplied key = the commitment of their
    interpolation over the commit key) now runs on sharded provers too — summed over the slices, or per rank for whole keys.
    Rank 1's key has two points swapped (both still valid subgroup points): every rank must fail with PLONK_ERR_DATA instead
    of producing proofs whose wire commitments do not verify."""
    log = _run_failing([sys.executable, "bench.py", "--gpus", "2", "--log-gates", "12", "--steps", "1", "--warmup", "0", "--no-extras"],
                       {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1", "PLONK_BENCH_WIRE_SPLIT": split,
                        "PLONK_BENCH_CORRUPT_LAGRANGE": "1"})
    assert "not the Lagrange-basis form of this context's commit key" in log, log[-3000:]


def test_self_launcher_starts_the_ranks_on_the_gpu_path():
    """`python bench.py --gpus 2` with no launcher in front of it (how the driver starts the bench): two ranks, the
    single-GPU proof"""
    s = single(12, "dense")
    m = _run([sys.executable, "bench.py", "--gpus", "2", "--log-gates", "12", "--steps", "1", "--warmup", "0", "--no-extras"],
             {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1"})
    assert m["n_gpus"] == 2 and m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.parametrize("ranks,log_gates,profile", [(1, 13, "widgets"), (4, 12, "dense")])   # (2, 13): tests/test_gpu_standin_transport.py
def test_compiled_prover_matches_the_coefficient_form_prover(ranks, log_gates, profile):
    """plonk_compile (Compiler::preprocess on the device: gate columns + witness indices in) on 1 / 2 / 4 ranks — the
    VerifierKey commitments are sharded MSMs like every other commitment — against the single-GPU prover built
    from coefficient forms; bench.py --from-circuit also proves from the witness table on every rank
    (plonk_prover_prove_witnesses) and compares with the wire-column entry point."""
    s = single(log_gates, profile)
    if ranks == 1:
        m = _run([sys.executable, "bench.py", "--log-gates", str(log_gates), "--steps", "1", "--warmup", "0", "--profile", profile,
                  "--no-cpu-baseline", "--no-extras", "--from-circuit"])
    else:
        m = multi(ranks, log_gates, profile, {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1"}, ["--from-circuit"])
    assert "plonk_compile" in m["config"]["prover_built_by"] and "plonk_prover_create" in s["config"]["prover_built_by"]
    assert m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.parametrize("ranks,log_gates,profile", [(2, 13, "widgets")])   # (8, 13), (4, 16): through the device all-gather, test_gpu_standin_transport.py
def test_sharded_grand_product_matches_single_gpu(ranks, log_gates, profile):
    """PLONK_SHARD_Z=1: the permutation grand product of round 2 split over the ranks (each rank its n / W evaluation indices,
    range products exchanged, z evaluations all-gathered in place) — the default from 2^19 gates on, forced here on small
    circuits; =0 (the replicated grand product) is what the other tests of this file run at these sizes."""
    s = single(log_gates, profile)
    m = multi(ranks, log_gates, profile, {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1", "PLONK_SHARD_Z": "1"})
    assert m["n_gpus"] == ranks and m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.parametrize("ranks,log_gates,env", [(3, 13, {}), (2, 12, {"PLONK_SHARD_QUOTIENT": "0"})])
def test_msm_only_sharding_matches_single_gpu(ranks, log_gates, env):
    """other world sizes / PLONK_SHARD_QUOTIENT=0: only the MSMs are sharded (round-1 path)."""
    s = single(log_gates, "dense")
    e = {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1"}
    e.update(env)
    m = multi(ranks, log_gates, "dense", e)
    assert m["proof_blake2b"] == s["proof_blake2b"]


def test_default_backend_self_test_and_fallback():
    """`bench.py --gpus 2` as the driver launches it (RCCL inside the library).  On this 1-GPU box both ranks
    share the device, which RCCL refuses: the bring-up must notice, every rank must agree on the gloo
    transport, and the proof must still equal the single-GPU one.  (On a real multi-GPU node the same
    code path reports "collective": "rccl".)"""
    s = single(12, "dense")
    m = multi(2, 12, "dense", {"PLONK_BENCH_SHARE_GPU": "1"})
    assert m["config"]["collective"] in ("rccl", "gloo")
    assert m["proof_blake2b"] == s["proof_blake2b"]


def test_rccl_transport_single_rank_communicator():
    """plonk_comm_unique_id / plonk_comm_init / plonk_comm_selftest with world = 1: RCCL is found and loaded,
    the communicator comes up on the context's device and ncclAllGather / ncclAllToAll run on the library's
    stream; a second init on the same context is refused; after destroy the context is reusable."""
    import plonk_amd
    ctx = plonk_amd.Context(0)
    uid = plonk_amd.Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(uid, 0, 1)
    ctx.comm_selftest()
    with pytest.raises(plonk_amd.PlonkError):
        ctx.comm_init(uid, 0, 1)
    ctx.comm_destroy()
    with pytest.raises(plonk_amd.PlonkError):
        ctx.comm_selftest()
    assert ctx.ntt([1, 2, 3, 4], 2) is not None
    ctx.close()


def test_comm_api_misuse_is_refused():
    import plonk_amd
    ctx = plonk_amd.Context(0)
    with pytest.raises(plonk_amd.PlonkError):          # no communicator yet
        ctx.comm_selftest()
    uid = plonk_amd.Context.comm_unique_id()
    with pytest.raises(plonk_amd.PlonkError):          # rank outside the world
        ctx.comm_init(uid, 2, 2)
    ctx.comm_destroy()                                 # destroying nothing is fine
    # a sharded prover without a communicator and without an all-gather callback cannot be built
    ctx.srs_load_bytes(bytes(96) * 0 + b"", 0)
    with pytest.raises(plonk_amd.PlonkError):
        plonk_amd.Prover(ctx, 64, b"x", {}, None, 0, 2, 71, None)
    ctx.close()


@pytest.mark.gpu
def test_rank_alone_measurement_tool_runs_and_ends_in_the_identity_check():
    """tools/rank_alone.py (plonk_comm_measure_loopback: a rank's own contribution in its peers' places) — the per-rank times of
    DESIGN.md section 5 come from it.  Every sharded proof made that way must fail at the FINAL quotient-identity check
    (PLONK_ERR_UNSAT), i.e. after all of the rank's work; the tool asserts that and never calls the transport callback."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rank_alone.py"), "12", "2", "2,8"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [json.loads(x) for x in r.stdout.strip().splitlines()]
    assert [x["world"] for x in lines] == [2, 8]
    assert all(x["callback_calls"] == 0 and x["prove_ms_rank_alone"] > 0 for x in lines)
    assert lines[0]["points"] == (4096 + 7 + 1) // 2
