import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# generated bench circuits are written once per session and shared by every test and every rank of every child process
# (bench_circuits._cached); a fresh directory per session, removed at exit
import atexit  # noqa: E402
import shutil  # noqa: E402
import tempfile  # noqa: E402

if "PLONK_CIRCUIT_CACHE" not in os.environ:
    try:
        _cache_dir = tempfile.mkdtemp(prefix="plonk_circuits_")
        os.environ["PLONK_CIRCUIT_CACHE"] = _cache_dir
        atexit.register(shutil.rmtree, _cache_dir, True)
    except OSError:      # no writable temporary directory: the suite runs without the cache (slower, same results)
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: longer CPU oracle case")


def configure(ctx, **fields):
    """plonk_ctx_set_config on top of the context's effective configuration (the library no longer reads switches from the
    environment after a context exists: tests that used monkeypatch.setenv on a live context now say what they mean)"""
    g = ctx.get_config()
    for k, v in fields.items():
        assert hasattr(g, k), k
        setattr(g, k, v)
    ctx.set_config(g)


@pytest.fixture(scope="session")
def kat_setup():
    """SRS + compiled MinimalCircuit of the reference KAT (prover.rs:1132-1147)."""
    from oracle.plonk import Composer, compile_circuit, srs_setup
    from oracle.rng import StdRng

    pp = srs_setup(1 << 10, StdRng.seed_from_u64(0x9235E700), keep=23)

    def circuit():
        c = Composer()
        w = c.append_witness(7)
        c.assert_equal_constant(w, 7)
        return c

    prover = compile_circuit(pp, b"proof-compatibility", circuit())
    return pp, prover, circuit
