"""The known-tau verifier (oracle/verifier.py, restating reference proof.rs:218-507) accepts
the oracle's own proofs — including the reference KAT proof — and rejects tampered ones."""
import random

from oracle import bls12_381 as E
from oracle import plonk as O
from oracle.rng import StdRng
from oracle.verifier import verify_with_tau

Q = E.Q


def srs_with_trapdoor(seed, max_degree, keep):
    rng = StdRng.seed_from_u64(seed)
    pp = O.srs_setup(max_degree, rng, keep=keep)
    replay = StdRng.seed_from_u64(seed)
    tau = replay.random_nonzero_scalar()
    return pp, tau


def vk_points(prover):
    return {k: v for k, v in prover.vk.items() if k != "n"}


def test_reference_kat_proof_verifies(kat_setup):
    _, prover, circuit = kat_setup
    proof, _ = O.prove(prover, StdRng.seed_from_u64(0x9235E701), circuit())
    _, tau = srs_with_trapdoor(0x9235E700, 1 << 10, 1)
    assert verify_with_tau(proof, vk_points(prover), prover.label, prover.constraints, {}, tau, prover.ck[0])
    bad = bytearray(proof)
    bad[600] ^= 1                                    # a_w evaluation
    assert not verify_with_tau(bytes(bad), vk_points(prover), prover.label, prover.constraints, {}, tau, prover.ck[0])
    assert not verify_with_tau(proof, vk_points(prover), b"other-label", prover.constraints, {}, tau, prover.ck[0])
    assert not verify_with_tau(proof, vk_points(prover), prover.label, prover.constraints, {}, tau + 1, prover.ck[0])


def test_random_circuit_with_public_inputs_verifies():
    from tests.test_gpu_prover import arithmetic_circuit, widget_circuit
    pp, tau = srs_with_trapdoor(77, 300, 80)
    for build in (arithmetic_circuit(40, 5), widget_circuit):
        prover = O.compile_circuit(pp, b"verify-me", build(), msm=E.msm_pippenger)
        comp = build()
        proof, pis = O.prove(prover, StdRng.seed_from_u64(9), comp, msm=E.msm_pippenger)
        pi = dict(comp.public_inputs)
        assert verify_with_tau(proof, vk_points(prover), prover.label, prover.constraints, pi, tau, prover.ck[0])
        if pi:
            k = next(iter(pi))
            wrong = dict(pi)
            wrong[k] = (wrong[k] + 1) % Q
            assert not verify_with_tau(proof, vk_points(prover), prover.label, prover.constraints, wrong, tau, prover.ck[0])
        swapped = proof[48:96] + proof[:48] + proof[96:]
        assert not verify_with_tau(swapped, vk_points(prover), prover.label, prover.constraints, pi, tau, prover.ck[0])
