"""examples/prove_from_blob.c: the C-ABI used from plain C (C99, no Python / torch in the process).
CPU: it compiles against include/plonk_hip.h, links against libplonk_hip.so, refuses a malformed blob on the host and
fails loudly without a GPU.  GPU: the C program reproduces the reference KAT digest (prover.rs:1151-1158)."""
import hashlib
import os
import subprocess

import pytest

import plonk_amd
from oracle.rng import StdRng
from oracle.serialize import prover_to_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if not os.path.exists(plonk_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build_hip(verbose=False)
    out = str(tmp_path_factory.mktemp("cex") / "prove_from_blob")
    libdir = os.path.dirname(plonk_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "prove_from_blob.c"), "-L" + libdir, "-lplonk_hip",
                           "-Wl,-rpath," + libdir, "-o", out])
    return out


@pytest.fixture(scope="module")
def kat_files(tmp_path_factory, kat_setup):
    from test_gpu_prover import wires_of
    _, op, circuit = kat_setup
    d = tmp_path_factory.mktemp("kat")
    rng = StdRng.seed_from_u64(0x9235E701)
    blinders = [rng.random_scalar() for _ in range(14)]
    cols = wires_of(circuit(), op.size)
    paths = {k: str(d / k) for k in ("blob", "wires", "blinders", "proof")}
    open(paths["blob"], "wb").write(prover_to_bytes(op))
    open(paths["wires"], "wb").write(b"".join(plonk_amd.fr_to_bytes_mont(c) for c in cols))
    open(paths["blinders"], "wb").write(plonk_amd.fr_to_bytes_mont(blinders))
    return paths


def test_usage_and_host_side_refusals(exe, kat_files, tmp_path):
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    bad = tmp_path / "bad.blob"
    bad.write_bytes(open(kat_files["blob"], "rb").read()[:-5])          # truncated: NotEnoughBytes before any device is touched
    r = subprocess.run([exe, str(bad), kat_files["wires"], kat_files["blinders"], str(tmp_path / "p")], capture_output=True, text=True)
    assert r.returncode == 1 and "plonk_prover_blob_check failed: code -8" in r.stderr
    short = tmp_path / "short.wires"
    short.write_bytes(b"\0" * 32)
    r = subprocess.run([exe, kat_files["blob"], str(short), kat_files["blinders"], str(tmp_path / "p")], capture_output=True, text=True)
    assert r.returncode == 2 and "4 x 8 x 32" in r.stderr


def test_without_a_gpu_the_c_program_fails_loudly(exe, kat_files):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe, kat_files["blob"], kat_files["wires"], kat_files["blinders"], kat_files["proof"]], capture_output=True, text=True)
    assert r.returncode == 1 and "plonk_ctx_create failed" in r.stderr
    assert not os.path.exists(kat_files["proof"])


@pytest.mark.gpu
def test_c_program_reproduces_the_reference_kat_digest(exe, kat_files):
    from test_gpu_prover import KAT_DIGEST
    r = subprocess.run([exe, kat_files["blob"], kat_files["wires"], kat_files["blinders"], kat_files["proof"]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    proof = open(kat_files["proof"], "rb").read()
    assert len(proof) == 1008 and hashlib.blake2b(proof).digest() == KAT_DIGEST
