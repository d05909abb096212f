"""GPU parity of plonk_compile (Compiler::preprocess on the device, reference src/compiler.rs:132-461) and
plonk_prover_prove_witnesses (prove_inner's wire columns gathered from the witness values, prover.rs:446-460):

  * the reference KAT through the compiled prover: MinimalCircuit laid out as gate columns -> VerifierKey
    commitments of the oracle's Compiler::preprocess -> blake2b(proof) == the literal at prover.rs:1151-1158;
  * circuits with every widget family at 2^12 / 2^13 gates: the 15 key polynomials coefficient by coefficient
    against the C oracle's interpolation of the same columns (tests/circuits.py compile_fast), the 15
    commitments against the C oracle prover's, the proof bytes against the C oracle's prove();
  * the error surface."""
import hashlib

import pytest

from oracle import bls12_381 as E
from oracle import cbind
from oracle.rng import StdRng
from tests import circuits as C
from tests.test_oracle_kat import KAT_DIGEST

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import plonk_amd
    c = plonk_amd.Context(0)
    yield c
    c.close()


def compiled(ctx, comp, label, **kw):
    import plonk_amd
    cols = C.circuit_columns(comp)
    return plonk_amd.Prover.compile(ctx, label, cols["selectors"], cols["wires"], cols["witnesses"], **kw), cols


@pytest.mark.parametrize("domain", ["quotient-4n", "quotient-8n"])
def test_kat_circuit_compiled_on_the_device(ctx, kat_setup, monkeypatch, domain):
    import plonk_amd
    from conftest import configure
    configure(ctx, quotient_domain=8 if domain == "quotient-8n" else 4)
    _, oprover, circuit = kat_setup
    ctx.srs_load(oprover.ck)
    comp = circuit()
    gp, cols = compiled(ctx, comp, b"proof-compatibility")
    assert gp.size == oprover.size == 8
    assert gp.vk_commitments() == b"".join(E.g1_compress(oprover.vk[n]) for n in plonk_amd.POLY_ORDER)
    for k, name in enumerate(plonk_amd.POLY_ORDER):
        want = list(oprover.pk.polys[name])
        got = gp.peek(12, k * (gp.size + 8), gp.size)
        assert got == want + [0] * (gp.size - len(want)), name
    rng = StdRng.seed_from_u64(0x9235E701)
    blinders = [rng.random_scalar() for _ in range(14)]
    proof = gp.prove_witnesses(cols["values"], {}, blinders)
    assert hashlib.blake2b(proof).digest() == KAT_DIGEST
    gp.close()


@pytest.mark.parametrize("log_n", [12, 13])
def test_compiled_key_and_proof_equal_the_c_oracle(ctx, log_n):
    import plonk_amd
    comp = C.big_widget_circuit(1 << log_n, seed=300 + log_n)()
    case = C.compile_fast(comp, b"compile-parity")
    n = case["size"]
    srs = C.synthetic_srs(n + 7)
    ctx.srs_load_bytes(srs, len(srs) // 96)
    gp, cols = compiled(ctx, comp, b"compile-parity")
    assert gp.size == n
    # key polynomials: iNTT of the same columns by the C oracle (EvaluationDomain::ifft)
    for k, name in enumerate(plonk_amd.POLY_ORDER):
        want = C.fr_vals(case["polys"][name]) if case["polys"][name] else [0] * n
        assert gp.peek(12, k * (n + 8), n) == want, name
    cp = cbind.CProver(case["constraints"], case["label"], case["polys"], srs)
    assert gp.vk_commitments() == cp.vk()
    bl = C.blinders(8100 + log_n)
    expected = cp.prove(case["wires"], case["pi_idx"], case["pi_val"], bl)
    got = gp.prove_witnesses(cols["values"], case["pi"], bl)
    assert got == expected
    # the gathered columns are the padded wire columns: the column entry point gives the same proof
    assert gp.prove(case["wires"], case["pi"], C.fr_vals(bl)) == expected
    # an unsatisfying witness table
    bad = bytearray(cols["values"])
    bad[32 * comp.constraints[len(comp.constraints) // 2].c] ^= 1
    with pytest.raises(plonk_amd.CircuitUnsatisfied):
        gp.prove_witnesses(bytes(bad), case["pi"], bl)
    assert gp.prove_witnesses(cols["values"], case["pi"], bl) == expected
    gp.close()


def test_compile_error_surface(ctx):
    import plonk_amd
    comp = C.big_widget_circuit(200, seed=5)()
    case = C.compile_fast(comp, b"errors")
    srs = C.synthetic_srs(case["size"] + 7)
    ctx.srs_load_bytes(srs, len(srs) // 96)
    cols = C.circuit_columns(comp)
    with pytest.raises(plonk_amd.PlonkError) as e:            # witness index out of range
        plonk_amd.Prover.compile(ctx, b"errors", cols["selectors"], cols["wires"], cols["witnesses"] - 1)
    assert e.value.code == -1
    gp = plonk_amd.Prover.compile(ctx, b"errors", cols["selectors"], cols["wires"], cols["witnesses"])
    with pytest.raises(plonk_amd.PlonkError) as e:            # wrong number of witness values
        gp.prove_witnesses(cols["values"][:-32], case["pi"], C.blinders(1))
    assert e.value.code == -1
    gp.close()
    plain = plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"])
    with pytest.raises(plonk_amd.PlonkError) as e:            # not a compiled prover
        plain.prove_witnesses(cols["values"], case["pi"], C.blinders(1))
    assert e.value.code == -7
    plain.close()
    ctx.srs_load_bytes(srs[:96 * 64], 64)                     # commit key shorter than the domain
    with pytest.raises(plonk_amd.PlonkError) as e:
        plonk_amd.Prover.compile(ctx, b"errors", cols["selectors"], cols["wires"], cols["witnesses"])
    assert e.value.code == -3


def test_cached_lagrange_key_gives_the_same_prover(ctx):
    """plonk_prover_desc.lagrange_xy96 on one GPU: a key kept from plonk_lagrange_key replaces the group FFT of the
    build; same commitments, same proof.  A key of the wrong length is refused."""
    import plonk_amd
    comp = C.big_widget_circuit(1 << 10, seed=77)()
    case = C.compile_fast(comp, b"cached-key")
    n, log_n = case["size"], case["log_n"]
    srs = C.synthetic_srs(n + 7)
    ctx.srs_load_bytes(srs, len(srs) // 96)
    key = ctx.lagrange_key(log_n)
    assert len(key) == 96 * (n + 2)
    cols = C.circuit_columns(comp)
    bl = C.blinders(41)
    fresh = plonk_amd.Prover.compile(ctx, b"cached-key", cols["selectors"], cols["wires"], cols["witnesses"])
    want = fresh.prove_witnesses(cols["values"], case["pi"], bl)
    fresh.close()
    for make in (lambda: plonk_amd.Prover.compile(ctx, b"cached-key", cols["selectors"], cols["wires"], cols["witnesses"], lagrange_slice=key),
                 lambda: plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], lagrange_slice=key)):
        gp = make()
        assert gp.prove(case["wires"], case["pi"], C.fr_vals(bl)) == want
        gp.close()
    with pytest.raises(plonk_amd.PlonkError) as e:
        plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], lagrange_slice=key[:-96])
    assert e.value.code == -1
    # a supplied key is checked against the context's commit key (a stale cached key used to yield proofs that
    # only the verifier rejected): two points swapped -> valid points, wrong key: PLONK_ERR_DATA
    bad = bytearray(key)
    bad[96 * 5:96 * 6], bad[96 * 6:96 * 7] = key[96 * 6:96 * 7], key[96 * 5:96 * 6]
    with pytest.raises(plonk_amd.PlonkError) as e:
        plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], lagrange_slice=bytes(bad))
    assert e.value.code == -9
    # the key of ANOTHER setup (tau + 1): PLONK_ERR_DATA as well
    other = C.synthetic_srs(n + 7, tau=(0x5EED0000 * 0x9E3779B97F4A7C15 + 1) % C.Q)
    ctx.srs_load_bytes(other, len(other) // 96)
    stale = ctx.lagrange_key(log_n)
    ctx.srs_load_bytes(srs, len(srs) // 96)
    with pytest.raises(plonk_amd.PlonkError) as e:
        plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], lagrange_slice=stale)
    assert e.value.code == -9
    # a point off the curve: PLONK_ERR_POINT
    bad = bytearray(key)
    bad[96 * 9] ^= 1
    with pytest.raises(plonk_amd.PlonkError) as e:
        plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], lagrange_slice=bytes(bad))
    assert e.value.code == -10
    # and the honest key still works afterwards
    gp = plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], lagrange_slice=key)
    assert gp.prove(case["wires"], case["pi"], C.fr_vals(bl)) == want
    gp.close()


def test_compiled_prover_serialises_like_the_reference(ctx, kat_setup):
    """plonk_prover_to_bytes on the KAT circuit compiled by the device: byte for byte the blob the oracle's restatement
    of Prover::to_bytes (oracle/serialize.py) writes for the oracle's Compiler::preprocess — its digest is the
    literal of tests/test_prover_blob.py that tools/dump_kat_blob.rs pins to the reference — and
    plonk_prover_from_bytes takes it back (the KAT proof digest again)."""
    import plonk_amd
    from oracle.serialize import prover_to_bytes
    _, oprover, circuit = kat_setup
    ctx.srs_load(oprover.ck)
    gp, cols = compiled(ctx, circuit(), b"proof-compatibility")
    blob = gp.to_bytes()
    gp.close()
    want = prover_to_bytes(oprover)
    assert len(blob) == len(want) == 43966
    assert blob == want
    assert hashlib.blake2b(blob).hexdigest().startswith("959ac0e3ee3d8f14695fccf849c92c9e")
    back = plonk_amd.Prover.from_bytes(ctx, blob)
    rng = StdRng.seed_from_u64(0x9235E701)
    blinders = [rng.random_scalar() for _ in range(14)]
    wires = C.wires_of(circuit(), 8)
    assert hashlib.blake2b(back.prove(wires, {}, blinders)).digest() == KAT_DIGEST
    back.close()


def test_serialised_prover_and_verifier_of_a_widget_circuit(ctx):
    """A 200-gate circuit with every widget family: plonk_prover_to_bytes against the oracle serialiser fed with the
    oracle's own compile (big-int interpolation, 8n coset evaluations, commitments); plonk_verifier_to_bytes against
    the restated Verifier::to_bytes; both blobs survive a round trip."""
    import plonk_amd
    from oracle import plonk as O
    from oracle.serialize import prover_to_bytes, verifier_to_bytes
    comp = C.big_widget_circuit(200, seed=11)()
    pp = O.srs_setup(300, StdRng.seed_from_u64(5), keep=256 + 7)
    oprover = O.compile_circuit(pp, b"serialise", C.big_widget_circuit(200, seed=11)(), msm=E.msm_pippenger)
    ctx.srs_load(oprover.ck)
    gp, cols = compiled(ctx, comp, b"serialise")
    blob = gp.to_bytes()
    assert blob == prover_to_bytes(oprover)
    info = plonk_amd.prover_blob_check(blob)
    assert info["size"] == 256 and info["constraints"] == len(comp.constraints)
    opening_key = bytes(range(240))                       # opaque to the library (G1 + 2 x G2 compressed)
    pi_idx = sorted(comp.public_inputs)
    vblob = gp.verifier_to_bytes(opening_key, pi_idx)
    assert vblob == verifier_to_bytes(b"serialise", oprover.vk, opening_key, pi_idx, 256, len(comp.constraints))
    bl = C.blinders(3)
    want = gp.prove_witnesses(cols["values"], dict(comp.public_inputs), bl)
    gp.close()
    back = plonk_amd.Prover.from_bytes(ctx, blob)
    case = C.compile_fast(comp, b"serialise")
    assert back.prove(case["wires"], case["pi"], C.fr_vals(bl)) == want
    assert back.to_bytes() == blob                        # and the loaded prover serialises to the same bytes
    back.close()
