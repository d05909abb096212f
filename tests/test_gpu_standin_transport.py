"""GPU, several ranks on ONE device, exchanges INSIDE the library: the RCCL branch of comm.hip (ncclAllGather through the
staging buffers, the in-place device all-gather of z, ncclAllToAll of the quotient remainders, comm_sync's polling) driven
with real peers through the stand-in transport of tests/fake_rccl (hipIpc device-pointer collectives between processes that
share the GPU, stream-ordered like RCCL's; RCCL itself refuses two ranks per device).  Before round 5 that branch had only
ever run with a 1-rank communicator, where every offset is 0 and every peer is self.

The sharded proofs must be the single-GPU proof byte for byte (whose bytes other tests compare with the oracle), and the
bench line must say what carried the data: "stand-in:libfakerccl.so", never "rccl", with n_ranks_rccl = 0.
Reference sites being sharded: prover.rs:187-210, quotient_poly.rs:139-157, permutation.rs:213-294."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")

from test_gpu_multirank import _run, single  # noqa: E402  (same launcher helpers, same cached single-GPU digests)


def standin(ranks, log_gates, profile, env=None, extra=()):
    assert os.path.exists(FAKE), "tests/fake_rccl/libfakerccl.so is built by __graft_entry__.build()"
    e = {"PLONK_BENCH_SHARE_GPU": "1", "PLONK_BENCH_TRANSPORT_LIBRARY": FAKE}
    e.update(env or {})
    m = _run([sys.executable, "bench.py", "--gpus", str(ranks), "--log-gates", str(log_gates), "--steps", "1", "--warmup", "1",
              "--profile", profile, "--no-extras", *extra], e)
    assert m["n_gpus"] == ranks
    assert m["config"]["collective"] == "stand-in:libfakerccl.so" and m["config"]["n_ranks_rccl"] == 0
    return m


def test_standin_library_exports_what_comm_hip_resolves():
    import ctypes
    lib = ctypes.CDLL(FAKE)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclCommCount", "ncclCommUserRank",
                 "ncclAllGather", "ncclAllToAll", "ncclGetErrorString"):
        assert hasattr(lib, name), name


@pytest.mark.parametrize("ranks,log_gates,profile", [(2, 13, "widgets"), (4, 13, "widgets"), (8, 13, "widgets"), (4, 16, "dense"),
                                                      (2, 12, "bench-like")])
def test_sharded_quotient_prove_through_device_collectives(ranks, log_gates, profile):
    """world in {2, 4, 8}: MSM partial sums through comm_allgather_host's staging, the class-sharded quotient through
    comm_alltoall_dev, rounds 4-5 through three more small all-gathers — all as nccl* calls on the library's stream"""
    s = single(log_gates, profile)
    m = standin(ranks, log_gates, profile, {"PLONK_BENCH_WIRE_SPLIT": "range"})   # every commitment by point range (the by-column split: below)
    assert "residue class" in m["config"]["parallelism"] and "by whole column" not in m["config"]["parallelism"]
    assert m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.parametrize("ranks,log_gates,profile", [(2, 13, "widgets"), (8, 13, "widgets"), (4, 16, "dense")])
def test_sharded_grand_product_through_the_in_place_all_gather(ranks, log_gates, profile):
    """PLONK_SHARD_Z=1: comm_allgather_dev with send = buf + rank * bytes, recv = buf (comm.hip) between real peers"""
    s = single(log_gates, profile)
    m = standin(ranks, log_gates, profile, {"PLONK_SHARD_Z": "1"})
    assert m["proof_blake2b"] == s["proof_blake2b"]


def test_compiled_prover_through_device_collectives():
    """plonk_compile on 2 ranks: the VerifierKey commitments are sharded MSMs whose partial sums travel the same way"""
    s = single(13, "widgets")
    m = standin(2, 13, "widgets", extra=["--from-circuit"])
    assert m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.parametrize("ranks,log_gates,profile", [(2, 13, "widgets"), (4, 13, "widgets"), (4, 16, "dense"), (2, 16, "bench-like")])
def test_wire_group_split_by_commitment(ranks, log_gates, profile):
    """Round 5: ranks of 2 / 4 that hold the WHOLE Lagrange-basis key commit to whole wire columns (rank r of 2: columns 2r,
    2r + 1; of 4: column r) and contribute the identity for the others — prover.rs:187-210's four commitments split by
    commitment instead of by point range.  Same proof bytes."""
    s = single(log_gates, profile)
    m = standin(ranks, log_gates, profile, {"PLONK_BENCH_WIRE_SPLIT": "commitment"})
    assert "by whole column" in m["config"]["parallelism"]
    assert m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.slow
def test_wire_group_split_by_point_range_at_2p20():
    """the round-4 split (a point range of every wire column per rank) at BASELINE config 4's size; the by-column split is what
    test_sharded_prove_at_2p20_through_device_collectives[2] and the gloo W = 4 case now run by default"""
    s = single(20, "dense")
    m = standin(2, 20, "dense", {"PLONK_BENCH_WIRE_SPLIT": "range"})
    assert "by whole column" not in m["config"]["parallelism"] and m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.slow
@pytest.mark.parametrize("ranks", [2, 8])
def test_sharded_prove_at_2p20_through_device_collectives(ranks):
    """BASELINE config 4's size: 2^20 gates over W = 2 (Q = 4) and W = 8 (Q = 8; sharded grand product by default) with the
    4 MiB-per-peer all-to-all and the 4 MiB z slices going through device pointers"""
    s = single(20, "dense")
    m = standin(ranks, 20, "dense")
    assert ("by whole column" in m["config"]["parallelism"]) == (ranks == 2)     # default for 2 / 4 ranks since round 5
    assert m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.slow
def test_sharded_prove_at_2p22_through_device_collectives():
    """BASELINE config 5's shape: 2^22 gates over W = 8 ranks (Q = 8 classes; 2^19 + 1 commit-key points per rank, each rank
    streaming only its own range from pinned host memory; 16 MiB per peer in the all-to-all, 16 MiB z slices in the in-place
    all-gather).  The eight ranks share this box's one GPU; the sharded proof must be the single-GPU proof byte for byte —
    whose bytes tests/test_gpu_fullsize.py compares with the C oracle."""
    s = single(22, "dense")
    m = standin(8, 22, "dense", extra=["--warmup", "0"])
    assert m["proof_blake2b"] == s["proof_blake2b"]


@pytest.mark.parametrize("die_at", ["alltoall:1", "allgather:2"])   # counted from the start of the first proof
def test_a_rank_that_dies_inside_a_proof_fails_the_others_within_the_timeout(die_at):
    """The ADVICE-r3 scenario, now executable: rank 1 of 2 exits on entering a collective of the first proof (the quotient
    all-to-all; an MSM all-gather).  Rank 0 is left inside that collective: it must return PLONK_ERR_STATE (-7) after about
    PLONK_COMM_TIMEOUT_MS (4 s here: poll, ncclCommAbort, bounded drain) — not hang — and refuse the next proof at once."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl", "peer_death.py"), "2", "12", "1", die_at],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    dead = [x for x in rep["ranks"] if x["rank"] == 1][0]
    alive = [x for x in rep["ranks"] if x["rank"] == 0][0]
    assert dead["exit"] == 17, rep
    assert alive["exit"] == 0 and alive["rc"] == -7, rep
    assert 3.0 <= alive["seconds"] <= 30.0, rep
    assert alive["rc_second"] == -7 and alive["seconds_second"] < 2.0, rep


def test_a_poisoned_context_refuses_every_entry_point_and_tears_down_without_hanging():
    """the transport's abort FAILS (fault injection in the stand-in), so after the time-out the dead collective's
    kernel still holds the survivor's stream and comm_sync marks the context unusable.  Until round 6 only the provers and
    plonk_comm_init checked that flag — plonk_msm / plonk_ntt / copies queued behind the dead kernel and blocked, and the destroy
    calls sat in hipStreamSynchronize.  Now: every such entry point returns PLONK_ERR_STATE at once, and prover + context
    teardown returns after the bounded polls (2 s each) with the device side abandoned."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl", "peer_death.py"), "2", "12", "1", "alltoall:1", "noabort"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    alive = [x for x in rep["ranks"] if x["rank"] == 0][0]
    assert alive["exit"] == 0 and alive["rc"] == -7, rep
    assert "unusable" in alive["error"], rep
    assert alive["rc_second"] == -7 and alive["seconds_second"] < 2.0, rep
    for name in ("msm", "ntt", "sync", "h2d"):
        assert alive["rc_" + name] == -7 and alive["seconds_" + name] < 1.0, (name, rep)
    assert alive["seconds_teardown"] < 12.0, rep      # three bounded polls of 2 s (prover, communicator, context) + the frees that are skipped


@pytest.mark.parametrize("ranks,log_gates,env", [(3, 13, {}), (2, 12, {"PLONK_SHARD_QUOTIENT": "0"})])
def test_msm_only_sharding_through_device_collectives(ranks, log_gates, env):
    """world sizes other than 2 / 4 / 8 (and PLONK_SHARD_QUOTIENT=0): only the MSMs are sharded — every exchange is one of
    comm_allgather_host's small staged all-gathers, here with an ODD number of peers"""
    s = single(log_gates, "dense")
    m = standin(ranks, log_gates, "dense", env)
    assert "residue class" not in m["config"]["parallelism"] or env
    assert m["proof_blake2b"] == s["proof_blake2b"]
