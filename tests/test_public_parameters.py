"""PublicParameters::to_raw_var_bytes (reference src/commitment_scheme/kzg10/srs.rs:114-146) -> plonk_public_parameters_check /
plonk_srs_load_public_parameters: the file a dusk-plonk user keeps next to the circuit.

The bytes come from the oracle's restatement of the serialiser (oracle/serialize.py) over the reference KAT's commit key
(SRS seed 0x9235_e700, prover.rs:1134-1139); the host-side decoder must find every piece and refuse what
PublicParameters::from_slice / from_slice_unchecked, OpeningKey::from_slice, CommitKey::from_raw_var_bytes and
CommitKey::truncate refuse, with the same error kinds.  The G2 half of the opening key is opaque to a prover backend."""
import pytest

import plonk_amd
from oracle import bls12_381 as E
from oracle.serialize import public_parameters_to_raw_var_bytes, public_parameters_to_var_bytes


def opening_key_bytes():
    # OpeningKey::to_bytes (key.rs:436-452): g (G1 generator), h (G2 generator), x_h = [x] h — real points: the loader decodes
    # all three like OpeningKey::from_slice (tests/g2_ref.py is the Python G2 used to make them)
    import g2_ref
    return E.g1_compress(E.G1_GEN) + g2_ref.g2_compress(g2_ref.G2_GEN) + g2_ref.g2_compress(g2_ref.g2_mul(g2_ref.G2_GEN, 0x1234567))


@pytest.fixture(scope="module")
def pp(kat_setup):
    _, oprover, _ = kat_setup
    return public_parameters_to_raw_var_bytes(opening_key_bytes(), oprover.ck), oprover.ck


def test_layout_and_trim(pp):
    data, ck = pp
    assert len(data) == 240 + 8 + 97 * len(ck)                                   # srs.rs:114-121, key.rs:215-229
    info = plonk_amd.public_parameters_check(data)
    assert info["opening_key"] == data[:240] and info["points_off"] == 248
    assert info["points_total"] == info["points_kept"] == len(ck) == 23
    off = info["points_off"]
    assert E.g1_from_raw96(data[off:off + 96]) == ck[0]
    assert E.g1_from_raw96(data[off + 97 * 22:off + 97 * 22 + 96]) == ck[22]
    # PublicParameters::trim(d) = CommitKey::truncate(d + 6): powers_of_g[..= d + 6]  (srs.rs:188-196, key.rs:336-355)
    assert plonk_amd.public_parameters_check(data, truncated_degree=8)["points_kept"] == 15
    assert plonk_amd.public_parameters_check(data, truncated_degree=16)["points_kept"] == 23   # max_degree = 22 = 16 + 6
    with pytest.raises(plonk_amd.PlonkError) as e:                               # Error::TruncatedDegreeTooLarge
        plonk_amd.public_parameters_check(data, truncated_degree=17)
    assert e.value.code == -3                                              # PLONK_ERR_DEGREE


def test_not_enough_bytes(pp):
    data, _ = pp
    for cut in (0, 100, 240):                                                    # srs.rs:165-167: len <= OpeningKey::SIZE
        with pytest.raises(plonk_amd.NotEnoughBytes):
            plonk_amd.public_parameters_check(data[:cut])
    with pytest.raises(plonk_amd.NotEnoughBytes):                                # commit key header
        plonk_amd.public_parameters_check(data[:244])
    with pytest.raises(plonk_amd.NotEnoughBytes):                                # from_raw_var_bytes: exact length
        plonk_amd.public_parameters_check(data[:-1])
    with pytest.raises(plonk_amd.NotEnoughBytes):
        plonk_amd.public_parameters_check(data + b"\x00")


def test_unchecked_decoding_takes_the_whole_chunks_present(pp):
    data, ck = pp
    # CommitKey::from_slice_unchecked (key.rs:243-258): chunks_exact(97).zip(0..count)
    info = plonk_amd.public_parameters_check(data[:-1], validate=False)
    assert info["points_total"] == len(ck) - 1
    info = plonk_amd.public_parameters_check(data + b"\x00" * 50, validate=False)
    assert info["points_total"] == len(ck)
    with pytest.raises(plonk_amd.NotEnoughBytes):
        plonk_amd.public_parameters_check(data[:248 + 96], validate=False)


def test_invalid_data(pp):
    data, _ = pp
    empty = data[:240] + (0).to_bytes(8, "little")
    with pytest.raises(plonk_amd.InvalidData):                                   # key.rs:272-274: len == 0
        plonk_amd.public_parameters_check(empty)
    bad_g = bytes([data[0] & 0x7F]) + data[1:]                                   # g without the compression flag
    with pytest.raises(plonk_amd.InvalidData):
        plonk_amd.public_parameters_check(bad_g)
    off_curve = E.g1_compress(E.G1_GEN)
    off_curve = off_curve[:47] + bytes([off_curve[47] ^ 1])                      # x + 1: x^3 + 4 is not a square here, or the point leaves the subgroup
    with pytest.raises(plonk_amd.InvalidData):
        plonk_amd.public_parameters_check(off_curve + data[48:])
    bad_h = data[:48] + bytes([data[48] & 0x7F]) + data[49:]
    with pytest.raises(plonk_amd.InvalidData):
        plonk_amd.public_parameters_check(bad_h)


def test_opening_key_g2_points_are_decoded_like_the_reference(pp):
    """OpeningKey::from_bytes (key.rs:596-615, then try_new :617-648) runs G2Affine::from_bytes on h and x_h: an encoding the reference refuses is
    refused here (they used to be checked for the compression flag only)."""
    import g2_ref
    data, _ = pp
    for off in (48, 144):                                                        # h, x_h
        enc = data[off:off + 96]
        assert g2_ref.g2_decompress(enc) is not None
        # x + 1 in c0: off the twist curve or outside the order-q subgroup — invalid either way
        bumped = bytearray(enc)
        bumped[95] ^= 1
        with pytest.raises(ValueError):
            g2_ref.g2_decompress(bytes(bumped))
        with pytest.raises(plonk_amd.InvalidData):
            plonk_amd.public_parameters_check(data[:off] + bytes(bumped) + data[off + 96:])
        # non-canonical coordinate (c0 = p)
        noncanon = enc[:48] + E.P.to_bytes(48, "big")
        with pytest.raises(plonk_amd.InvalidData):
            plonk_amd.public_parameters_check(data[:off] + noncanon + data[off + 96:])
        # the other root is the negated point: still a valid key
        neg = bytes([enc[0] ^ 0x20]) + enc[1:]
        assert plonk_amd.public_parameters_check(data[:off] + neg + data[off + 96:])["points_total"] == 23
        # the identity is a valid G2Affine encoding but OpeningKey::try_new (key.rs:617-648) refuses it: InvalidData
        ident = bytes([0xC0]) + bytes(95)
        with pytest.raises(plonk_amd.InvalidData):
            plonk_amd.public_parameters_check(data[:off] + ident + data[off + 96:])
    # the same for g (a valid compressed G1 identity)
    with pytest.raises(plonk_amd.InvalidData):
        plonk_amd.public_parameters_check(bytes([0xC0]) + bytes(47) + data[48:])


def test_point_malformed(pp):
    data, _ = pp
    inf = bytearray(data)
    inf[248 + 97 * 3 + 96] = 1                                                   # infinity flag of point 3
    with pytest.raises(plonk_amd.PointMalformed):
        plonk_amd.public_parameters_check(bytes(inf))
    big = bytearray(data)
    big[248 + 97 * 5 + 47] = 0xFF                                                # x limb 11 >= p's: not reduced
    with pytest.raises(plonk_amd.PointMalformed):
        plonk_amd.public_parameters_check(bytes(big))
    # CommitKey::from_raw_var_bytes tests every point BEFORE the trim (key.rs:263-300, srs.rs:188-196): a bad point beyond
    # the kept prefix refuses the file; from_slice_unchecked never looks there
    far = bytearray(data)
    far[248 + 97 * 20 + 47] = 0xFF
    with pytest.raises(plonk_amd.PointMalformed):
        plonk_amd.public_parameters_check(bytes(far), truncated_degree=8)
    assert plonk_amd.public_parameters_check(bytes(far), truncated_degree=8, validate=False)["points_kept"] == 15


def test_divergence_identity_in_the_commit_key_is_refused(pp):
    """DIVERGENCE from the reference, on purpose (include/plonk_hip.h): G1Affine::from_bytes and is_on_curve & is_torsion_free
    accept the identity, so dusk-plonk loads a commit key holding one; this library answers PLONK_ERR_POINT in every mode —
    no SRS [tau^i] G contains the identity and the precomputed table rows cannot represent it."""
    data, ck = pp
    raw = bytearray(data)
    raw[248 + 97 * 3:248 + 97 * 4] = bytes(96) + b"\x01"
    for validate in (True, False):
        with pytest.raises(plonk_amd.PointMalformed):
            plonk_amd.public_parameters_check(bytes(raw), validate=validate)
    comp = public_parameters_to_var_bytes(opening_key_bytes(), ck)
    comp = comp[:240 + 48 * 3] + bytes([0xC0]) + bytes(47) + comp[240 + 48 * 4:]
    with pytest.raises(plonk_amd.PointMalformed):
        plonk_amd.public_parameters_check(comp, compressed=True)


def test_compressed_form_layout_and_host_side_refusals(pp):
    _, ck = pp
    data = public_parameters_to_var_bytes(opening_key_bytes(), ck)                # srs.rs:149-153, key.rs:303-308
    assert len(data) == 240 + 48 * len(ck)
    info = plonk_amd.public_parameters_check(data, compressed=True)
    assert info["points_off"] == 240 and info["point_stride"] == 48 and info["points_total"] == info["points_kept"] == 23
    assert E.g1_decompress(data[240:288]) == ck[0]
    assert plonk_amd.public_parameters_check(data, truncated_degree=8, compressed=True)["points_kept"] == 15
    with pytest.raises(plonk_amd.PlonkError) as e:
        plonk_amd.public_parameters_check(data, truncated_degree=17, compressed=True)
    assert e.value.code == -3
    with pytest.raises(plonk_amd.NotEnoughBytes):                                # srs.rs:165-167
        plonk_amd.public_parameters_check(data[:240], compressed=True)
    with pytest.raises(plonk_amd.InvalidData):                                   # a short last chunk: G1Affine::from_slice fails
        plonk_amd.public_parameters_check(data[:-1], compressed=True)
    noflag = bytearray(data)
    noflag[240 + 48 * 2] &= 0x7F
    with pytest.raises(plonk_amd.InvalidData):
        plonk_amd.public_parameters_check(bytes(noflag), compressed=True)
    ident = data[:240 + 48 * 4] + bytes([0xC0]) + bytes(47) + data[240 + 48 * 5:]
    with pytest.raises(plonk_amd.PointMalformed):                                # valid encoding, but no commit key holds the identity
        plonk_amd.public_parameters_check(ident, compressed=True)
    badident = data[:240 + 48 * 4] + bytes([0xC0]) + bytes(46) + b"\x01" + data[240 + 48 * 5:]
    with pytest.raises(plonk_amd.InvalidData):
        plonk_amd.public_parameters_check(badident, compressed=True)


@pytest.mark.gpu
def test_compressed_key_is_decompressed_on_the_device(pp):
    _, ck = pp
    data = public_parameters_to_var_bytes(opening_key_bytes(), ck)
    ctx = plonk_amd.Context(0)
    ok = ctx.srs_load_public_parameters(data, compressed=True)                   # square roots + subgroup test on the GPU
    assert ok == data[:240] and ctx.srs_points == 23
    sc = [(0xA24BAED4963EE407 * (i + 5)) % E.Q for i in range(23)]
    assert ctx.msm(sc) == E.msm_pippenger(ck, sc)
    ctx.srs_load_public_parameters(data, truncated_degree=8, compressed=True)
    assert ctx.srs_points == 15
    assert ctx.msm(sc[:15]) == E.msm_pippenger(ck[:15], sc[:15])
    # both roots decode to the point that was encoded
    flipped = [(x, (E.P - y) % E.P) for x, y in ck]
    ctx.srs_load_public_parameters(public_parameters_to_var_bytes(opening_key_bytes(), flipped), compressed=True)
    assert ctx.msm(sc) == E.msm_pippenger(flipped, sc)
    # x^3 + 4 not a square -> InvalidData from the device pass; a point of the curve outside the subgroup likewise
    x = next(v for v in range(2, 50) if pow((v ** 3 + 4) % E.P, (E.P - 1) // 2, E.P) != 1)
    bad = data[:240 + 48 * 3] + bytes([0x80]) + x.to_bytes(47, "big") + data[240 + 48 * 4:]
    with pytest.raises(plonk_amd.InvalidData):
        ctx.srs_load_public_parameters(bad, compressed=True)
    x = next(v for v in range(2, 50) if pow((v ** 3 + 4) % E.P, (E.P - 1) // 2, E.P) == 1)   # on the curve, cofactor part present
    bad = data[:240 + 48 * 3] + bytes([0x80]) + x.to_bytes(47, "big") + data[240 + 48 * 4:]
    with pytest.raises(plonk_amd.InvalidData):
        ctx.srs_load_public_parameters(bad, compressed=True)
    ctx.close()


@pytest.mark.gpu
def test_loaded_key_commits_like_the_points_loaded_directly(pp):
    data, ck = pp
    ctx = plonk_amd.Context(0)
    ok = ctx.srs_load_public_parameters(data, truncated_degree=8)                # validate: on-curve + torsion-free on the GPU
    assert ok == data[:240] and ctx.srs_points == 15
    sc = [(0x9E3779B97F4A7C15 * (i + 1)) % E.Q for i in range(15)]
    assert ctx.msm(sc) == E.msm_pippenger(ck[:15], sc)
    with pytest.raises(plonk_amd.PolynomialDegreeTooLarge):                      # the trimmed key holds 15 points
        ctx.commit(list(range(1, 17)))
    ctx.srs_load_public_parameters(data, validate=False)                         # from_slice_unchecked: all 23 points
    assert ctx.srs_points == 23
    sc = [(0xD1B54A32D192ED03 * (i + 3)) % E.Q for i in range(23)]
    assert ctx.msm(sc) == E.msm_pippenger(ck, sc)
    # a point off the curve is caught by the GPU check of the validated path only
    bad = bytearray(data)
    bad[248 + 97 * 2 + 48] ^= 1                                                  # y of point 2
    with pytest.raises(plonk_amd.PointMalformed):
        ctx.srs_load_public_parameters(bytes(bad))
    ctx.close()


@pytest.mark.gpu
def test_validated_modes_test_every_point_of_the_file_before_the_trim(pp):
    """key.rs:263-300 / :319-326 decode and test the whole commit key, PublicParameters::trim cuts afterwards: a point off the
    curve BEYOND the kept prefix refuses the file in the validating modes (only the prefix used to be tested) and
    is never seen by from_slice_unchecked."""
    data, ck = pp
    ctx = plonk_amd.Context(0)
    bad = bytearray(data)
    bad[248 + 97 * 20 + 48] ^= 1                                                 # y of point 20; trim(8) keeps 15 points
    with pytest.raises(plonk_amd.PointMalformed):
        ctx.srs_load_public_parameters(bytes(bad), truncated_degree=8)
    ctx.srs_load_public_parameters(bytes(bad), truncated_degree=8, validate=False)
    assert ctx.srs_points == 15
    comp = public_parameters_to_var_bytes(opening_key_bytes(), ck)
    x = next(v for v in range(2, 50) if pow((v ** 3 + 4) % E.P, (E.P - 1) // 2, E.P) != 1)
    badc = comp[:240 + 48 * 20] + bytes([0x80]) + x.to_bytes(47, "big") + comp[240 + 48 * 21:]
    with pytest.raises(plonk_amd.InvalidData):
        ctx.srs_load_public_parameters(badc, truncated_degree=8, compressed=True)
    ctx.srs_load_public_parameters(comp, truncated_degree=8, compressed=True)
    assert ctx.srs_points == 15
    ctx.close()
