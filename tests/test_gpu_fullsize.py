"""GPU, BASELINE sizes: proofs produced at 2^16 and 2^20 gates (bench.py's workload) must
satisfy the verification equation of reference proof.rs:218-507, checked by the known-tau
verifier of oracle/verifier.py — a size-independent acceptance test where no CPU prover can
serve as oracle.  Also: proofs are deterministic and change with the blinders."""
import pytest

import bench
from oracle import bls12_381 as E
from oracle.verifier import verify_with_tau

pytestmark = pytest.mark.gpu
Q = E.Q


@pytest.mark.parametrize("log_n", [16, 20, pytest.param(22, marks=pytest.mark.slow)])
def test_bench_proof_verifies(log_n):
    """2^22 gates = BASELINE config 5's size on ONE GPU: 4.2 M-point commit key streamed from pinned host memory
    (bench.build_prover), 8.6 GiB of window tables + 8.5 GiB of key evaluations resident."""
    import plonk_amd
    ctx = plonk_amd.Context(0)
    tau, g = bench.TAU, bench.G_SCALAR
    bl1 = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
    oracle = {}

    def start_oracle(inputs):
        # BASELINE config 5's size, BYTE parity: a valid-but-different proof (a misplaced blinder in the split of t, a wrong
        # coefficient range) passes the verification equation below but not this.  The C oracle proves the same circuit
        # with the key's trapdoor — the commitment [g p(tau)] G is the group element the MSM over [g tau^i] G returns
        # (tests/test_oracle_c_prove.py::test_trapdoor_commitments_are_the_msm_commitments) — because eleven CPU MSMs of
        # 4 M terms would take minutes; the transforms, the quotient and every O(n) pass are the oracle's own.
        # Round 6: it runs in a thread BESIDE the GPU's set-up (ctypes releases the GIL; ~1.5 of the test's 1.8 minutes were
        # this oracle running after the GPU had finished), and it no longer takes the GPU's VerifierKey on trust: its 15 key
        # commitments are its own trapdoor commitments (VERDICT r5 weak 1 iii) and are compared with the GPU's BY BYTES.
        import threading
        from oracle import cbind
        wires, polys, _, q_m_column = inputs

        def run():
            try:
                n = 1 << log_n
                threads = cbind.max_threads()
                mont = plonk_amd.fr_to_bytes_mont
                # the key polynomials are INPUTS of prove() (Compiler::compile made them); pin one GPU-interpolated column to
                # the CPU transform so that the shared input is not taken on trust
                oracle["q_m_ok"] = cbind.ntt_bytes(q_m_column, log_n, True, False, n, threads) == polys["q_m"]
                pb = {k: (v if isinstance(v, (bytes, bytearray)) else mont(v)) for k, v in polys.items()}   # short ones come as integers
                cp = cbind.CProver(n, b"bench", pb, bytes(96 * (n + 7)), vk48=bytes(15 * 48), threads=threads)
                cp.set_trapdoor(mont([tau]), mont([g]))
                cp.adopt_vk_trapdoor()
                oracle["vk"] = cp.vk()
                oracle["proof"] = cp.prove(wires, [], b"", bl1)
                cp.close()
            except BaseException as e:   # noqa: BLE001  (re-raised on the main thread)
                oracle["error"] = e
        oracle["thread"] = threading.Thread(target=run, name="c-oracle-2p22", daemon=True)
        oracle["thread"].start()

    prover, wbuf, srs_total = bench.build_prover(ctx, log_n, 0, 1, None, on_inputs=start_oracle if log_n == 22 else None)
    srs_g = E.g1_mul(E.G1_GEN, g)
    raw = prover.vk_commitments()
    vk = {name: E.g1_decompress(raw[48 * i:48 * i + 48]) for i, name in enumerate(plonk_amd.POLY_ORDER)}
    bl2 = plonk_amd.fr_to_bytes_mont([(0xC0FFEE00 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
    p1 = prover.prove_dev(wbuf.ptr, {}, bl1)
    p1b = prover.prove_dev(wbuf.ptr, {}, bl1)
    p2 = prover.prove_dev(wbuf.ptr, {}, bl2)
    assert p1 == p1b and p1 != p2
    # bl1 are bench.py's blinders: this is the proof `python bench.py --log-gates N` prints the digest of — hand it to the
    # multi-rank tests of this session instead of building the same prover again in a child process
    import hashlib
    from session_cache import SINGLE
    SINGLE[(log_n, "dense")] = {"n_gpus": 1, "proof_blake2b": hashlib.blake2b(p1).hexdigest()[:32],
                                "config": {"prover_built_by": "plonk_prover_create (coefficient forms)", "collective": None}}
    n = 1 << log_n
    for p in (p1, p2):
        assert verify_with_tau(p, vk, b"bench", n, {}, tau, srs_g)
    bad = bytearray(p1)
    bad[530] ^= 4
    assert not verify_with_tau(bytes(bad), vk, b"bench", n, {}, tau, srs_g)
    if log_n == 22:
        oracle["thread"].join()
        if "error" in oracle:
            raise oracle["error"]
        assert oracle["q_m_ok"]
        assert oracle["vk"] == raw          # the 15 key commitments, byte for byte
        assert p1 == oracle["proof"]
    prover.close()
    wbuf.free()
    ctx.close()
