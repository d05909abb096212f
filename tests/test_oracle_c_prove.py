"""Pins the C restatement of the WHOLE prove() (oracle/c/oracle_prove.c) — CPU only:
  * it reproduces the reference's KAT digest (prover.rs:1151-1158);
  * it equals the big-int oracle byte for byte on circuits with every widget family active and
    public inputs (the oracle that is itself pinned to the KAT);
  * its SRS generator equals the big-int setup; its error codes mirror the reference's.
It is the checker of the GPU parity tests at 2^12 .. 2^20 gates (tests/test_gpu_prove_sizes.py)."""
import hashlib

import pytest

from oracle import bls12_381 as E
from oracle import cbind
from oracle import plonk as O
from oracle.rng import StdRng
from tests import circuits as C
from tests.test_oracle_kat import KAT_DIGEST
from tests.widget_circuits import semantic_widget_circuit

Q = E.Q


class Recorder:
    def __init__(self, rng):
        self.rng, self.drawn = rng, []

    def random_scalar(self):
        s = self.rng.random_scalar()
        self.drawn.append(s)
        return s


def c_prover_from(oprover):
    polys = {k: C.fr_bytes(v) for k, v in oprover.pk.polys.items()}
    srs = b"".join(E.g1_to_raw96(p) for p in oprover.ck)
    return cbind.CProver(oprover.constraints, oprover.label, polys, srs)


def test_c_prove_reproduces_reference_kat_digest(kat_setup):
    _, oprover, circuit = kat_setup
    cp = c_prover_from(oprover)
    assert cp.vk() == b"".join(E.g1_compress(oprover.vk[n]) for n in cbind.POLY_ORDER)
    rng = StdRng.seed_from_u64(0x9235E701)
    bl = C.fr_bytes([rng.random_scalar() for _ in range(14)])
    comp = circuit()
    proof = cp.prove([C.fr_bytes(w) for w in C.wires_of(comp, oprover.size)], [], b"", bl)
    assert hashlib.blake2b(proof).digest() == KAT_DIGEST


@pytest.mark.parametrize("seed", [1, 2])
def test_c_prove_matches_bigint_oracle_with_widgets_and_public_inputs(seed):
    pp = O.srs_setup(300, StdRng.seed_from_u64(91), keep=300)
    build = semantic_widget_circuit(seed)
    oprover = O.compile_circuit(pp, b"c-parity", build(), msm=E.msm_pippenger)
    rec = Recorder(StdRng.seed_from_u64(500 + seed))
    comp = build()
    trace = {}
    expected, pis = O.prove(oprover, rec, comp, msm=E.msm_pippenger, trace=trace)
    assert len(pis) == 2
    cp = c_prover_from(oprover)
    idx = sorted(comp.public_inputs)
    got, st = cp.prove([C.fr_bytes(w) for w in C.wires_of(comp, oprover.size)], idx,
                       C.fr_bytes([comp.public_inputs[i] for i in idx]), C.fr_bytes(rec.drawn), trace=True)
    assert got == expected
    n = oprover.size
    zp = C.fr_vals(st["z_poly"])
    assert zp[:n + 3] == trace["z_poly"] and not any(zp[n + 3:])
    # the fast column compiler of tests/circuits.py produces the same key polynomials
    fast = C.compile_fast(build(), b"c-parity")
    for name in cbind.POLY_ORDER:
        want = oprover.pk.polys[name]
        have = C.fr_vals(fast["polys"][name])
        assert have[:len(want)] == want and not any(have[len(want):]), name


def test_legacy_v2_transcript_differs_only_in_the_seeding(kat_setup):
    """prove_with_version(V2) (prover.rs:365-413, feature `legacy-proving`): Transcript::base + seed_transcript_legacy bind the label
    s_sigma_4 to the commitment of s_sigma_1 (widget.rs:224-228,260-265).  Both restatements agree byte for byte on a V2 proof,
    it differs from the V3 proof (whose digest is the reference's KAT), and the commitments of round 1 — made before the first
    challenge is drawn — are the same in both.  The reference holds no literal for V2: this pins the two restatements to each
    other and to the code read, not to reference-produced bytes."""
    _, oprover, circuit = kat_setup
    cp = c_prover_from(oprover)
    rng = StdRng.seed_from_u64(0x9235E701)
    bl = [rng.random_scalar() for _ in range(14)]
    comp = circuit()
    wires = [C.fr_bytes(w) for w in C.wires_of(comp, oprover.size)]
    v3 = cp.prove(wires, [], b"", C.fr_bytes(bl))
    cp.set_version(2)
    v2 = cp.prove(wires, [], b"", C.fr_bytes(bl))
    cp.set_version(3)
    assert cp.prove(wires, [], b"", C.fr_bytes(bl)) == v3 and hashlib.blake2b(v3).digest() == KAT_DIGEST
    assert v2 != v3 and v2[:4 * 48] == v3[:4 * 48]

    class Replay:
        def __init__(self, vals):
            self.vals = list(vals)

        def random_scalar(self):
            return self.vals.pop(0)
    big, _ = O.prove(oprover, Replay(bl), circuit(), version=2)
    assert big == v2


def test_c_prove_rejects_unsatisfied_witness_like_the_reference():
    """quotient_poly.rs:132 -> Error::CircuitUnsatisfied (tests/common/mod.rs:60-80)."""
    case = C.compile_fast(C.big_widget_circuit(128, seed=3)(), b"unsat")
    srs = b"".join(E.g1_to_raw96(p) for p in O.srs_setup(140, StdRng.seed_from_u64(5), keep=140))
    cp = cbind.CProver(case["constraints"], case["label"], case["polys"], srs)
    good = cp.prove(case["wires"], case["pi_idx"], case["pi_val"], C.blinders(1))
    assert len(good) == 1008
    bad = bytearray(case["wires"][2])
    bad[32 * 9] ^= 1
    with pytest.raises(cbind.CircuitUnsatisfied):
        cp.prove([case["wires"][0], case["wires"][1], bytes(bad), case["wires"][3]], case["pi_idx"], case["pi_val"], C.blinders(1))


def test_c_srs_generator_matches_bigint_setup():
    tau, g = 0x1234567 * 0x9E3779B97F4A7C15 % Q, 0xDEADBEEF
    raw = cbind.srs_generate(C.fr_bytes([tau]), C.fr_bytes([g]), 12)
    base = E.g1_mul(E.G1_GEN, g)
    for i in (0, 1, 2, 11):
        assert E.g1_from_raw96(raw[96 * i:96 * i + 96]) == E.g1_mul(base, pow(tau, i, Q))


def test_trapdoor_commitments_are_the_msm_commitments():
    """oracle_prover_set_trapdoor: on a synthetic key [g tau^i] G the commitment [g p(tau)] G is the group element the
    MSM returns — same proof bytes with and without it (the 2^22-gate GPU parity test relies on this mode)."""
    case = C.compile_fast(C.big_widget_circuit(512, seed=8)(), b"trapdoor")
    tau, g = 0x5EED0000 * 0x9E3779B97F4A7C15 % Q, 0xA5A5A5A5DEADBEEF
    srs = C.synthetic_srs(case["size"] + 7, tau, g)
    cp = cbind.CProver(case["constraints"], case["label"], case["polys"], srs)
    want = cp.prove(case["wires"], case["pi_idx"], case["pi_val"], C.blinders(31))
    cp.set_trapdoor(C.fr_bytes([tau]), C.fr_bytes([g]))
    assert cp.prove(case["wires"], case["pi_idx"], case["pi_val"], C.blinders(31)) == want
    assert cp.vk_trapdoor() == cp.vk()                            # the 15 key commitments too (they were MSMs at construction)
    # the way the 2^22-gate GPU test builds its oracle: a placeholder VerifierKey and no key points at all, everything through the trapdoor
    blind = cbind.CProver(case["constraints"], case["label"], case["polys"], bytes(len(srs)), vk48=bytes(15 * 48))
    blind.set_trapdoor(C.fr_bytes([tau]), C.fr_bytes([g]))
    blind.adopt_vk_trapdoor()
    assert blind.vk() == cp.vk()
    assert blind.prove(case["wires"], case["pi_idx"], case["pi_val"], C.blinders(31)) == want
    blind.close()
    cp.set_trapdoor(C.fr_bytes([tau + 1]), C.fr_bytes([g]))      # a wrong trapdoor is a different proof
    assert cp.vk_trapdoor() != cp.vk()
    assert cp.prove(case["wires"], case["pi_idx"], case["pi_val"], C.blinders(31)) != want
    cp.close()
