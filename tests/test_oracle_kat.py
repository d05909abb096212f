"""Pins the oracle against the reference's own golden vectors."""
import hashlib

from oracle import bls12_381 as E
from oracle.fft import EvaluationDomain, serial_fft
from oracle.merlin import Transcript, keccak_f1600
from oracle.plonk import poly_ruffini, prove
from oracle.rng import StdRng

Q = E.Q

# reference src/compiler/prover.rs:1151-1158
KAT_DIGEST = bytes([
    0xe8, 0x56, 0x4e, 0xc2, 0x2d, 0x8c, 0xc0, 0xba, 0x60, 0x36, 0x26,
    0x02, 0x5d, 0xa3, 0x75, 0x50, 0x77, 0xaa, 0xf0, 0x32, 0x32, 0x61,
    0x90, 0x8d, 0xab, 0x68, 0xd6, 0x94, 0x73, 0x6f, 0xc2, 0x73, 0xd3,
    0x1e, 0x25, 0x6c, 0xbd, 0x3a, 0x6a, 0x21, 0xe7, 0xad, 0xe6, 0x31,
    0x91, 0xac, 0x5c, 0x9d, 0x44, 0xa1, 0x13, 0xac, 0x49, 0x89, 0xa5,
    0x2e, 0x4b, 0xe3, 0xab, 0xeb, 0x1d, 0x33, 0x32, 0x37,
])


def test_deterministic_v3_proof_matches_base_digest(kat_setup):
    """reference prover.rs:1132-1162 — the single end-to-end KAT."""
    _, prover, circuit = kat_setup
    assert (len(prover.ck), prover.size, prover.constraints) == (23, 8, 5)
    proof, pis = prove(prover, StdRng.seed_from_u64(0x9235E701), circuit())
    assert pis == []
    assert len(proof) == 1008
    assert hashlib.blake2b(proof).digest() == KAT_DIGEST


def test_minus_one_literal_pins_montgomery_form():
    """reference src/composer.rs:334-339 MINUS_ONE == (-R mod q) limbs."""
    assert E.fr_to_mont_limbs(Q - 1) == [0xfffffffd00000003, 0xfb38ec08fffb13fc,
                                         0x99ad88181ce5880f, 0x5bc8f5f97cd877d8]
    assert E.fr_from_mont_limbs(E.fr_to_mont_limbs(12345)) == 12345


def test_root_of_unity_order():
    assert pow(E.ROOT_OF_UNITY, 1 << 32, Q) == 1
    assert pow(E.ROOT_OF_UNITY, 1 << 31, Q) != 1


def test_keccak_matches_sha3():
    st = bytearray(200)
    st[0] ^= 0x06
    st[135] ^= 0x80
    keccak_f1600(st)
    assert bytes(st[:32]) == hashlib.sha3_256(b"").digest()


def test_merlin_is_deterministic_and_label_sensitive():
    a = Transcript(b"test protocol")
    a.append_message(b"some label", b"some data")
    b = Transcript(b"test protocol")
    b.append_message(b"some label", b"some data")
    c = Transcript(b"test protocol")
    c.append_message(b"some label", b"some datb")
    ca, cb, cc = (t.challenge_bytes(b"challenge", 32) for t in (a, b, c))
    assert ca == cb != cc


def test_linear_coset_evaluations_match_closed_form():
    """reference domain.rs:620-636: coset_fft([0,1]) on 2^8 == 7*w^i."""
    d = EvaluationDomain(1 << 8)
    ev = d.coset_fft([0, 1])
    exp, cur = [], E.GENERATOR
    for _ in range(d.size):
        exp.append(cur)
        cur = cur * d.group_gen % Q
    assert ev == exp


def test_vanishing_coset_evaluations_match_closed_form():
    """reference domain.rs:638-651."""
    d = EvaluationDomain(1 << 8)
    got = d.vanishing_poly_over_coset(32)
    roots = d.elements()
    assert got == [(pow(7 * r % Q, 32, Q) - 1) % Q for r in roots]


def test_fft_ifft_roundtrip_and_naive_dft():
    """reference domain.rs:570-618 input (i+1), smaller size; fft == naive DFT."""
    d = EvaluationDomain(64)
    a = [i + 1 for i in range(64)]
    ev = d.fft(a)
    naive = [sum(a[j] * pow(d.group_gen, i * j, Q) for j in range(64)) % Q for i in range(64)]
    assert ev == naive
    assert d.ifft(ev) == a
    assert d.coset_ifft(d.coset_fft(a)) == a


def test_fft_truncates_longer_input():
    """reference domain.rs:174 Vec::resize truncation."""
    d = EvaluationDomain(8)
    a = list(range(1, 13))
    assert d.fft(a) == d.fft(a[:8])


def test_serial_fft_rejects_wrong_length():
    import pytest
    with pytest.raises(AssertionError):
        serial_fft([1, 2, 3], 1, 2)


def test_ruffini_small():
    """(x^2 - 1) / (x - 1) = x + 1 (reference polynomial.rs:469-500 style)."""
    assert poly_ruffini([Q - 1, 0, 1], 1) == [1, 1]


def test_g1_small_linear_combination():
    """reference kzg10/proof.rs:120-158 style: 2G,3G,5G with v=7."""
    G = E.G1_GEN
    pts = [E.g1_mul(G, 2), E.g1_mul(G, 3), E.g1_mul(G, 5)]
    got = E.msm_naive(pts, [1, 7, 49])
    assert got == E.g1_mul(G, 2 + 21 + 245)
    assert E.msm_pippenger(pts, [1, 7, 49]) == got
    assert E.g1_decompress(E.g1_compress(got)) == got
    assert E.g1_compress(None) == bytes([0xC0]) + bytes(47)
    assert E.g1_from_raw96(E.g1_to_raw96(got)) == got


def test_pippenger_matches_naive_random():
    import random
    r = random.Random(5)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, Q)) for _ in range(40)]
    sc = [r.randrange(Q) for _ in range(40)]
    sc[3] = 0
    sc[4] = 1
    sc[5] = Q - 1
    assert E.msm_pippenger(pts, sc) == E.msm_naive(pts, sc)


MERLIN_SIMPLE = "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_merlin_published_vector():
    """merlin 3.0 `transcript::tests::equivalence_simple`: Transcript("test protocol"),
    append_message("some label", "some data"), challenge_bytes("challenge", 32) — an anchor for the
    STROBE-128 / Keccak restatement that is independent of the PLONK KAT."""
    t = Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == MERLIN_SIMPLE


def test_g1_generator_compressed_encoding():
    """The BLS12-381 G1 generator in the zcash 48-byte compressed form (G1Affine::to_bytes,
    commitment.rs:49-51) — the published constant, for the oracle and for the product-side encoder."""
    import plonk_amd
    want = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    assert E.g1_compress(E.G1_GEN).hex() == want
    assert plonk_amd.g1_compress(E.G1_GEN).hex() == want
    assert E.g1_decompress(bytes.fromhex(want)) == E.G1_GEN
    assert E.g1_compress(None) == bytes([0xC0]) + bytes(47)
