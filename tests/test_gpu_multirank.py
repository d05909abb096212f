"""GPU, 2 ranks on ONE device (gloo exchange): the point-range-sharded prove() of bench.py
must output the same Proof bytes as the single-GPU run.  Exercises prover.hip's msm_group /
fetch_commitments all-gather path and the ctypes exchange callback on real hardware (RCCL
itself cannot run two ranks on one device; the driver's 2/4/8-GPU runs use backend nccl)."""
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


@pytest.mark.parametrize("ranks,log_gates", [(2, 12), (3, 13)])
def test_sharded_prove_matches_single_gpu(ranks, log_gates):
    single = _run([sys.executable, "bench.py", "--log-gates", str(log_gates), "--steps", "1", "--warmup", "0",
                   "--no-cpu-baseline"])
    multi = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
                  "--master-addr", "127.0.0.1", "--master-port", str(29600 + ranks), "bench.py", "--gpus", str(ranks),
                  "--log-gates", str(log_gates), "--steps", "1", "--warmup", "0"],
                 {"PLONK_BENCH_BACKEND": "gloo", "PLONK_BENCH_SHARE_GPU": "1"})
    assert multi["n_gpus"] == ranks and single["n_gpus"] == 1
    assert multi["proof_blake2b"] == single["proof_blake2b"]


def test_default_backend_self_test_and_fallback():
    """`bench.py --gpus 2` as the driver launches it (backend nccl = RCCL).  On this 1-GPU box both
    ranks share the device, which RCCL refuses: the self-test must notice, every rank must agree
    on the gloo fallback, and the proof must still equal the single-GPU one.  (On a real multi-GPU
    node the same code path reports "collective": "rccl".)"""
    single = _run([sys.executable, "bench.py", "--log-gates", "12", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    multi = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                  "--master-addr", "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2",
                  "--log-gates", "12", "--steps", "1", "--warmup", "0"],
                 {"PLONK_BENCH_SHARE_GPU": "1"})
    assert multi["config"]["collective"] in ("rccl", "gloo")
    assert multi["proof_blake2b"] == single["proof_blake2b"]
