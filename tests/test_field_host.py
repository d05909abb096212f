"""Product field/curve code (plonk_amd/csrc/field.cuh, curve.cuh) compiled for the
HOST and compared bit for bit with the big-int oracle.  CPU-only."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import bls12_381 as E

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libhost_arith.so")
Q, P = E.Q, E.P


def build_host_lib():
    """g++ build of tests/csrc/host_arith.cpp: the device headers compiled for the host (HD macros)."""
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(HERE, "csrc", "host_arith.cpp")
    hdrs = [os.path.join(HERE, "..", "plonk_amd", "csrc", h)
            for h in ("field.cuh", "curve.cuh", "fp28.cuh", "curve28.cuh", "fr29.cuh", "transcript.hpp", "widgets.hpp", "hostg1.hpp", "permutation.hpp", "g1codec.cuh",
                      "msm_recode.cuh", "fp_safegcd.cuh", "hostg2.hpp", "api_guard.hpp", "finish_pool.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(f) > os.path.getmtime(SO) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", src, "-o", SO])
    return ctypes.CDLL(SO)


@pytest.fixture(scope="module")
def lib():
    return build_host_lib()


def fr_limbs(x):
    m = x * E.FR_R % Q
    return (ctypes.c_uint32 * 8)(*[(m >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def fr_val(buf):
    return sum(int(v) << (32 * i) for i, v in enumerate(buf)) * E.FR_RINV % Q


def fp_limbs(x):
    m = x * E.FP_R % P
    return (ctypes.c_uint32 * 12)(*[(m >> (32 * i)) & 0xFFFFFFFF for i in range(12)])


def fp_val(buf):
    return sum(int(v) << (32 * i) for i, v in enumerate(buf)) * E.FP_RINV % P


def edge_values(mod, rnd, n=60):
    vals = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << 32) - 1, 1 << 32, (1 << 255) % mod]
    return vals + [rnd.randrange(mod) for _ in range(n)]


def test_fr_ops_match_oracle(lib):
    rnd = random.Random(1)
    vals = edge_values(Q, rnd)
    out = (ctypes.c_uint32 * 8)()
    for a in vals:
        for b in vals[:12] + [rnd.randrange(Q)]:
            lib.h_fr_mul(fr_limbs(a), fr_limbs(b), out)
            assert fr_val(out) == a * b % Q
            lib.h_fr_add(fr_limbs(a), fr_limbs(b), out)
            assert fr_val(out) == (a + b) % Q
            lib.h_fr_sub(fr_limbs(a), fr_limbs(b), out)
            assert fr_val(out) == (a - b) % Q
    for a in vals[1:20]:
        lib.h_fr_inv(fr_limbs(a), out)
        assert fr_val(out) == pow(a, -1, Q)
    lib.h_fr_from_mont(fr_limbs(12345), out)
    assert sum(int(v) << (32 * i) for i, v in enumerate(out)) == 12345


def test_fr_results_are_canonical_limbs(lib):
    """Outputs must be fully reduced: identical limbs to the reference's BlsScalar.0."""
    out = (ctypes.c_uint32 * 8)()
    lib.h_fr_sub(fr_limbs(0), fr_limbs(1), out)
    limbs64 = [int(out[2 * i]) | int(out[2 * i + 1]) << 32 for i in range(4)]
    assert limbs64 == [0xfffffffd00000003, 0xfb38ec08fffb13fc, 0x99ad88181ce5880f, 0x5bc8f5f97cd877d8]


def test_fr_constants(lib):
    out = (ctypes.c_uint32 * 24)()
    lib.h_fr_consts(out)
    assert fr_val(out[0:8]) == 7
    assert fr_val(out[8:16]) == E.ROOT_OF_UNITY
    assert fr_val(out[16:24]) == 1


def test_fp_ops_match_oracle(lib):
    rnd = random.Random(2)
    vals = edge_values(P, rnd, 40)
    out = (ctypes.c_uint32 * 12)()
    for a in vals:
        for b in vals[:10] + [rnd.randrange(P)]:
            lib.h_fp_mul(fp_limbs(a), fp_limbs(b), out)
            assert fp_val(out) == a * b % P
            lib.h_fp_add(fp_limbs(a), fp_limbs(b), out)
            assert fp_val(out) == (a + b) % P
            lib.h_fp_sub(fp_limbs(a), fp_limbs(b), out)
            assert fp_val(out) == (a - b) % P
    for a in vals[1:8]:
        lib.h_fp_inv(fp_limbs(a), out)
        assert fp_val(out) == pow(a, -1, P)


def test_g1_group_law_matches_oracle(lib):
    rnd = random.Random(3)
    G = E.G1_GEN
    pts = [E.g1_mul(G, rnd.randrange(1, Q)) for _ in range(6)]
    out = (ctypes.c_uint8 * 96)()
    for a in pts:
        for b in pts:
            ra, rb = E.g1_to_raw96(a), E.g1_to_raw96(b)
            ok = lib.h_g1_add_aff(ra, rb, out)       # includes a == b (doubling branch)
            assert ok == 1 and E.g1_from_raw96(bytes(out)) == E.g1_add(a, b)
            ok = lib.h_g1_add_full(ra, rb, out)
            assert ok == 1 and E.g1_from_raw96(bytes(out)) == E.g1_add(a, b)
        assert lib.h_g1_neg_add(E.g1_to_raw96(a), out) == 0     # P + (-P) = identity
        for k in (1, 2, 3, 0xFFFF, 0x80000001):
            ok = lib.h_g1_mul_u32(E.g1_to_raw96(a), k, out)
            assert ok == 1 and E.g1_from_raw96(bytes(out)) == E.g1_mul(a, k)
        assert lib.h_g1_mul_u32(E.g1_to_raw96(a), 0, out) == 0


def test_fp28_reduced_radix_matches_oracle(lib):
    """fp28.cuh (14 x 28-bit limbs, lazy reduction) against big ints, through the 32-bit form."""
    rnd = random.Random(9)
    vals = edge_values(P, rnd, 60)
    out = (ctypes.c_uint32 * 12)()
    for a in vals:
        lib.h_fp28_roundtrip(fp_limbs(a), out)
        assert fp_val(out) == a
        for b in vals[:9] + [rnd.randrange(P), rnd.randrange(P)]:
            lib.h_fp28_mul(fp_limbs(a), fp_limbs(b), out)
            assert fp_val(out) == a * b % P
            lib.h_fp28_chain(fp_limbs(a), fp_limbs(b), out)
            assert fp_val(out) == a * b % P
        flags = lib.h_fp28_zero_test(fp_limbs(a))
        assert flags & 3 == 3 and bool(flags & 4) == (a == 0)


def test_fp28_lazy_operands_sqr_and_fused_product(lib):
    """fp28.cuh: dedicated squaring, mul2 (two products, one reduction) and lazy (un-normalised,
    limbs < 2^30) operands, checked on raw limbs incl. the all-ones patterns that maximise the
    64-bit column accumulators."""
    rnd = random.Random(28)
    M28 = (1 << 28) - 1
    RINV = pow(1 << 392, -1, P)
    A14 = ctypes.c_uint32 * 14

    def limbs(v):   # normalised 14 x 28
        return [(v >> (28 * i)) & M28 for i in range(14)]

    def val(ls):
        return sum(int(x) << (28 * i) for i, x in enumerate(ls))

    def run(op, a, b, c, d):
        out = A14()
        lib.h_fp28_raw(op, A14(*a), A14(*b), A14(*c), A14(*d), out)
        assert all(x <= M28 for x in out[:13]), "result not normalised"
        return val(out)

    def lazy(op, a, b):
        out = A14()
        lib.h_fp28_lazy(op, A14(*a), A14(*b), out)
        return list(out)

    ones = [M28] * 13 + [0x1a010]            # < p, every low limb saturated
    norm = [limbs(rnd.randrange(2 * P)) for _ in range(6)] + [ones, limbs(0), limbs(1), limbs(P - 1), limbs(2 * P - 1)]
    zero = limbs(0)
    for a in norm:
        for b in norm:
            # lazy minuend/subtrahend combos (values: a - b + 32p < 34p needs b < 16p: ok, b < 2p)
            la = lazy(0, a, b)
            assert max(la) < (1 << 30) and val(la) == val(a) + 32 * P - val(b)
            nb = lazy(1, zero, b)
            assert val(nb) == 16 * P - val(b)
            r = run(1, la, zero, zero, zero)                       # sqr of a lazy operand, value < 34p
            assert r < 2 * P and r % P == val(la) ** 2 * RINV % P
            r = run(0, la, la, zero, zero)                         # lazy x lazy through mul
            assert r < 2 * P and r % P == val(la) ** 2 * RINV % P
            r = run(1, a, zero, zero, zero)
            assert r < 2 * P and r % P == val(a) ** 2 * RINV % P
            # mul2 as the point formulas use it: (norm x lazy) + (norm x lazy)
            r = run(2, a, la, b, nb)
            assert r < 2 * P and r % P == (val(a) * val(la) + val(b) * val(nb)) * RINV % P
            dbl = lazy(2, b, b)
            assert val(dbl) == 2 * val(b)


def test_g1_xyzz_over_fp28_matches_oracle(lib):
    """curve28.cuh: lazily-reduced XYZZ formulas — long accumulation chains (bounds must stay
    closed), negated operands, P + P and P + (-P) through the mixed and the full addition."""
    rnd = random.Random(12)
    G = E.G1_GEN
    pts = [E.g1_mul(G, rnd.randrange(1, Q)) for _ in range(40)]
    raw = b"".join(E.g1_to_raw96(p) for p in pts)
    out = (ctypes.c_uint8 * 96)()
    for n in (1, 2, 3, 17, 40):
        neg = bytes(rnd.randrange(2) for _ in range(n))
        ok = lib.h_g1r_accumulate(raw, neg, n, out)
        exp = None
        for p, s in zip(pts[:n], neg):
            exp = E.g1_add(exp, (p[0], (E.P - p[1]) % E.P) if s else p)
        assert (ok == 1) == (exp is not None)
        if exp is not None:
            assert E.g1_from_raw96(bytes(out)) == exp
    # same point repeatedly: first mixed add hits the doubling branch, then generic adds
    rep = E.g1_to_raw96(pts[0]) * 9
    assert lib.h_g1r_accumulate(rep, bytes(9), 9, out) == 1
    assert E.g1_from_raw96(bytes(out)) == E.g1_mul(pts[0], 9)
    # P + (-P) + P
    assert lib.h_g1r_accumulate(rep, bytes([0, 1, 0]), 3, out) == 1
    assert E.g1_from_raw96(bytes(out)) == pts[0]
    assert lib.h_g1r_accumulate(rep, bytes([0, 1]), 2, out) == 0
    for k in (1, 2, 16 * 2047, 0xFFFF):
        ok = lib.h_g1r_tree(raw, 20, k, out)
        s = None
        for p in pts[:20]:
            s = E.g1_add(s, p)
        assert ok == 1 and E.g1_from_raw96(bytes(out)) == E.g1_mul(s, 2 * k)
    lib.h_g1r_affine_roundtrip(E.g1_to_raw96(pts[3]), out)
    assert E.g1_from_raw96(bytes(out)) == E.g1_mul(pts[3], 4)


def test_fr29_reduced_radix_butterflies_match_oracle(lib):
    """fr29.cuh: lazy DIF butterflies (a + b, (a - b) w) incl. 9 chained stages."""
    rnd = random.Random(29)
    vals = edge_values(Q, rnd, 40)
    o0, o1 = (ctypes.c_uint32 * 8)(), (ctypes.c_uint32 * 8)()
    for a in vals:
        for b in vals[:10] + [rnd.randrange(Q)]:
            w = rnd.choice(vals)
            lib.h_fr29_butterfly(fr_limbs(a), fr_limbs(b), fr_limbs(w), o0, o1)
            assert fr_val(o0) == (a + b) % Q and fr_val(o1) == (a - b) * w % Q
            # outputs must be canonical limbs
            assert sum(int(v) << (32 * i) for i, v in enumerate(o0)) < Q
            x, y = a, b
            for _ in range(9):
                x, y = (x + y) % Q, (x - y) * w % Q
            lib.h_fr29_chain(fr_limbs(a), fr_limbs(b), fr_limbs(w), 9, o0, o1)
            assert fr_val(o0) == x and fr_val(o1) == y
            lib.h_fr29_mul2(fr_limbs(a), fr_limbs(b), fr_limbs(w), o0)
            assert fr_val(o0) == a * b * w % Q


def test_fr29_sub_reduce(lib):
    rnd = random.Random(31)
    vals = edge_values(Q, rnd, 60)
    o = (ctypes.c_uint32 * 8)()
    for a in vals:
        for b in vals[:12] + [rnd.randrange(Q)]:
            lib.h_fr29_sub_reduce(fr_limbs(a), fr_limbs(b), o)
            assert fr_val(o) == (2 * a - 2 * b) % Q


def test_host_transcript_matches_merlin_vector(lib):
    """transcript.hpp (the C++ Merlin / STROBE-128 / Keccak-f[1600] used by prover.hip) on Merlin's
    published `equivalence_simple` vector."""
    from test_oracle_kat import MERLIN_SIMPLE
    out = (ctypes.c_uint8 * 32)()
    lib.h_merlin_simple(out)
    assert bytes(out).hex() == MERLIN_SIMPLE


def test_host_msm_finish_and_group_normalisation(lib):
    """hostg1.hpp: W = sum_j 2^j U_j from the 16 bit sums of msm_bits_kernel (rows T_0..T_7, columns
    T'_0..T'_6, C_128), the shared-inversion affine normalisation and the compressed encoding, in
    64-bit-limb host arithmetic, against the oracle's group law."""
    rnd = random.Random(64)
    G = E.G1_GEN
    for trial in range(12):
        bitpos = trial >= 4                       # last sum S: the result is 2 W - S (bit-position entries weigh 2 b + 1)
        rb = 12 if trial >= 8 else 8              # row bit sums: 8 for 2^15 buckets, 12 for 2^19
        pts = [E.g1_mul(G, rnd.randrange(1, Q)) for _ in range(rb + 8 + (1 if bitpos else 0))]
        if trial % 4 == 1:
            pts[3] = None                         # an empty bit sum
            pts[9] = None
        if trial % 4 == 2:
            pts = [None] * len(pts)               # all-zero polynomial: the commitment is the identity
        if trial % 4 == 3:
            pts = [pts[0]] * len(pts)             # equal points: the additions hit the doubling branch
        raw = b"".join(E.g1_to_raw96(p) if p is not None else bytes(96) for p in pts)
        out = (ctypes.c_uint8 * 48)()
        lib.h_finish_bit_sums(raw, rb, 1 if bitpos else 0, out)
        # rows weigh 2^(7+j) (indices 0..rb-1), columns 2^j (indices rb..rb+6), C_128 weighs 2^7 (index rb+7)
        want = None
        for k, p in enumerate(pts[:rb + 8]):
            if p is None:
                continue
            w = (1 << (7 + k)) if k < rb else ((1 << (k - rb)) if k < rb + 7 else (1 << 7))
            want = E.g1_add(want, E.g1_mul(p, w))
        if bitpos:
            want = E.g1_add(want, want)
            if pts[rb + 8] is not None:
                want = E.g1_add(want, E.g1_mul(pts[rb + 8], Q - 1))
        assert bytes(out) == E.g1_compress(want), trial
    pts = [E.g1_mul(G, rnd.randrange(1, Q)) for _ in range(15)]
    pts[4] = None
    raw = b"".join(E.g1_to_raw96(p) if p is not None else bytes(96) for p in pts)
    out = (ctypes.c_uint8 * (48 * 15))()
    lib.h_batch_compress(raw, 15, out)
    assert bytes(out) == b"".join(E.g1_compress(p) for p in pts)


def test_g1_decompress48_matches_the_oracle(lib):
    """g1codec.cuh (the lane function of the compressed commit-key loader) against oracle g1_compress / g1_decompress:
    G1Affine::from_bytes semantics of the 48-byte encoding — both roots, the flag rules, x >= p, x^3 + 4 not a square."""
    rnd = random.Random(11)
    out = ctypes.create_string_buffer(96)
    pts = [E.G1_GEN] + [E.g1_mul(E.G1_GEN, rnd.randrange(1, Q)) for _ in range(24)]
    pts += [(x, (P - y) % P) for x, y in pts[:8]]                      # the other root / sign flag
    for pt in pts:
        enc = E.g1_compress(pt)
        assert lib.h_g1_decompress48(enc, out) == 0
        assert out.raw == E.g1_to_raw96(pt)
        assert E.g1_decompress(enc) == pt
    g = E.g1_compress(E.G1_GEN)
    assert lib.h_g1_decompress48(bytes([g[0] & 0x7F]) + g[1:], out) == 1          # compression flag missing
    assert lib.h_g1_decompress48(bytes([0xC0]) + bytes(47), out) == 2              # the identity
    assert lib.h_g1_decompress48(bytes([0xE0]) + bytes(47), out) == 1              # identity with the sort flag
    assert lib.h_g1_decompress48(bytes([0xC0]) + bytes(46) + b"\x01", out) == 1    # identity with x != 0
    assert lib.h_g1_decompress48(bytes([0x80 | (P >> 376)]) + (P & ((1 << 376) - 1)).to_bytes(47, "big"), out) == 1   # x = p
    bad = 0
    for x in range(1, 40):                                                          # x^3 + 4 a non-residue for about half of them
        enc = bytes([0x80]) + x.to_bytes(47, "big")
        rc = lib.h_g1_decompress48(enc, out)
        on_curve = pow((x ** 3 + 4) % P, (P - 1) // 2, P) == 1
        assert rc == (0 if on_curve else 1)
        bad += not on_curve
        if on_curve:                                                                # smaller root requested (flag clear)
            y = pow((x ** 3 + 4) % P, (P + 1) // 4, P)
            y = min(y, P - y)
            assert out.raw == E.g1_to_raw96((x, y))
    assert bad > 5


def test_safegcd_inverse_matches_the_oracle(lib):
    """fp_safegcd.cuh (Bernstein-Yang division steps, 13 signed 30-bit limbs) through the Fp28 interface: x -> x^-1 for edge
    values and random ones, lazily reduced inputs included; 0 -> 0."""
    rnd = random.Random(381)
    out = (ctypes.c_uint32 * 12)()
    vals = edge_values(P, rnd, 150) + [(1 << k) % P for k in (1, 29, 30, 31, 59, 60, 380)] + [P - (1 << 30), (P + 1) // 2, 3, P - 3]
    for a in vals:
        lib.h_fp_inv_gcd(fp_limbs(a), out)
        assert fp_val(out) == (pow(a, -1, P) if a else 0), hex(a)
        lib.h_fp_inv_gcd_lazy(fp_limbs(a), out)
        assert fp_val(out) == (pow(8 * a % P, -1, P) if a else 0), hex(a)


def test_safegcd_inverse_in_fr_matches_the_oracle(lib):
    """the 9-limb instance (batch_inverse_kernel's workgroup inversion): x -> x^-1 mod q in twiddle form"""
    rnd = random.Random(255)
    out = (ctypes.c_uint32 * 8)()
    for a in edge_values(Q, rnd, 200) + [Q - (1 << 30), (Q + 1) // 2, 7, pow(7, (Q - 1) >> 32, Q)]:
        lib.h_fr_inv_gcd(fr_limbs(a), out)
        assert fr_val(out) == (pow(a, -1, Q) if a else 0), hex(a)


def test_g2_reference_generator_is_the_standard_one():
    """tests/g2_ref.py (the checker of the next test): generator on E'(Fp2), of order q, and its well-known encoding"""
    import g2_ref as G
    from oracle.bls12_381 import Q as ORDER
    assert G.on_curve(G.G2_GEN) and G.g2_mul(G.G2_GEN, ORDER) is None
    assert G.g2_compress(G.G2_GEN).hex().startswith("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049")
    assert G.g2_decompress(G.g2_compress(G.G2_GEN)) == G.G2_GEN


def test_g2_compressed_validity_matches_the_reference_decoder(lib):
    """hostg2.hpp g2_compressed_valid == "G2Affine::from_bytes succeeds" (key.rs:596-648 through dusk-bls12_381): multiples of
    the generator with either sign flag, the identity, and every way an encoding can be wrong — flag bits, a non-canonical
    coordinate, x off the curve, a curve point outside the order-q subgroup."""
    import g2_ref as G
    r = random.Random(77)

    def valid(b):
        return bool(lib.h_g2_compressed_valid(bytes(b)))

    def ref(b):
        try:
            G.g2_decompress(bytes(b))
            return True
        except ValueError:
            return False

    good = [G.g2_compress(G.g2_mul(G.G2_GEN, k)) for k in (1, 2, 3, r.randrange(1, 1 << 255), r.randrange(1, 1 << 255))]
    for enc in good:
        assert ref(enc) and valid(enc)
        flipped = bytearray(enc)
        flipped[0] ^= 0x20                                     # the other root: still a point of the subgroup
        assert ref(flipped) and valid(flipped)
        no_flag = bytearray(enc)
        no_flag[0] &= 0x7F
        assert not valid(no_flag)
        inf_flag = bytearray(enc)
        inf_flag[0] |= 0x40                                    # infinity flag on a finite x
        assert not ref(inf_flag) and not valid(inf_flag)
    identity = bytes([0xC0]) + bytes(95)
    assert ref(identity) and valid(identity)
    assert not valid(bytes([0xE0]) + bytes(95))               # identity with the sign flag
    assert not valid(bytes([0xC0]) + bytes(94) + b"\x01")     # identity with a stray bit
    # non-canonical: c0 = p (zero + p), c1 = p + small
    from oracle.bls12_381 import P as MODP
    enc = bytearray(good[0])
    enc[48:] = MODP.to_bytes(48, "big")
    assert not ref(enc) and not valid(enc)
    # x values by trial: off the curve, and on the curve but outside the subgroup (cofactor of E'(Fp2) is huge)
    seen_off = seen_outside = 0
    x1 = 1
    while seen_off < 3 or seen_outside < 3:
        x1 += 1
        x = (r.randrange(MODP), x1)
        b = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
        b[0] |= 0x80
        y = G.f2_sqrt(G.f2_add(G.f2_mul(G.f2_sqr(x), x), G.B2))
        if y is None:
            seen_off += 1
            assert not valid(b)
        else:
            assert G.g2_mul((x, y), 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001) is not None
            seen_outside += 1
            assert not ref(b) and not valid(b)


def test_g1_compressed_validity_on_host(lib):
    from oracle import bls12_381 as E
    assert lib.h_g1_compressed_valid(E.g1_compress(E.G1_GEN)) == 1
    assert lib.h_g1_compressed_valid(bytes([0xC0]) + bytes(47)) == 1
    bad = bytearray(E.g1_compress(E.G1_GEN))
    bad[0] &= 0x7F
    assert lib.h_g1_compressed_valid(bytes(bad)) == 0


def test_host_fp_inverse_safegcd_equals_fermat_and_the_oracle(lib):
    """hostg1.hpp fp64_inv (the shared inversion of a commitment group's affine normalisation, on the host between two GPU
    phases): Bernstein-Yang divsteps instead of the a^(p-2) chain — Montgomery form in and out, 0 -> 0."""
    r = random.Random(91)
    R = 1 << 384
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 380, (1 << 381) % P] + [r.randrange(P) for _ in range(200)]
    out = ctypes.create_string_buffer(48)
    for v in vals:
        mont = (v * R % P).to_bytes(48, "little")
        expect = (pow(v, -1, P) * R % P) if v else 0
        for fermat in (0, 1):
            lib.h_fp64_inv(mont, fermat, out)
            assert int.from_bytes(out.raw, "little") == expect, (v, fermat)


def test_host_fr_inverse_safegcd_montgomery_form(lib):
    """fp_safegcd.cuh fr_inv_gcd: x R -> x^-1 R on the 8 x 32-bit Fr — the per-proof host inversions of prover.hip
    (1 / (z (z - 1)), the public-input denominators); 0 -> 0 like Fr::inv."""
    r = random.Random(92)
    out = (ctypes.c_uint32 * 8)()
    for v in [0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, 1 << 254] + [r.randrange(Q) for _ in range(300)]:
        lib.h_fr_inv_gcd_mont(fr_limbs(v), out)
        assert fr_val(out) == (pow(v, -1, Q) if v else 0), v


def test_abi_exception_barrier_turns_exceptions_into_codes(lib):
    """plonk_amd/csrc/api_guard.hpp: every int-returning entry point of the C-ABI runs inside api_guard — a C++ exception
    never unwinds into the C / Rust caller (include/plonk_hip.h: PLONK_ERR_NOMEM for std::bad_alloc, else PLONK_ERR_STATE)."""
    msg = ctypes.create_string_buffer(256)
    assert lib.h_api_guard(0, msg) == 7 and msg.value == b""
    assert lib.h_api_guard(1, msg) == -11 and b"bad_alloc" in msg.value and msg.value.startswith(b"h_api_guard")
    assert lib.h_api_guard(2, msg) == -7 and b"boom" in msg.value
    assert lib.h_api_guard(3, msg) == -7 and b"unknown" in msg.value


def test_every_int_entry_point_runs_inside_the_exception_barrier():
    """Source check: each `int plonk_*` definition of the library opens with the api_guard lambda (the one-line
    plonk_ctx_table_rows cannot throw)."""
    import re
    root = os.path.join(HERE, "..", "plonk_amd", "csrc")
    seen = 0
    for f in ("capi.hip", "prover.hip", "serial.hip", "comm.hip"):
        lines = open(os.path.join(root, f)).read().split("\n")
        for i, line in enumerate(lines):
            m = re.match(r"^int (plonk_\w+)\(", line)
            if not m or m.group(1) == "plonk_ctx_table_rows":
                continue
            j = i
            while not lines[j].rstrip().endswith(("{", ";")):
                j += 1
            if lines[j].rstrip().endswith(";"):
                continue                                   # a declaration
            assert "api_guard(api_fn" in lines[j + 2], (f, m.group(1))
            seen += 1
    assert seen >= 42


def test_cooperative_16_lane_product_model(lib):
    """tools/ubench/coop_mul.hpp (measurement prototype): the 16-lane cooperative Montgomery product
    with its lanes emulated by arrays — columns from shifted operands, redundant quotient, the two-pass carry into column 14
    and the lane-13 flag — equals a b / 2^392 modulo p for normalised, lazy and extreme operands, with limbs under 2^30."""
    r = random.Random(1628)
    RP = 1 << 392
    rinv = pow(RP, -1, P)

    def limbs(x, lazy=0):
        ls = [(x >> (28 * i)) & 0xFFFFFFF for i in range(14)]
        for i in range(lazy):                        # the same value with some limbs pushed above 2^28 (borrowing from the next)
            j = r.randrange(13)
            if ls[j + 1] > 0 and ls[j] + (1 << 28) < (1 << 30):
                ls[j + 1] -= 1
                ls[j] += 1 << 28
        return ls

    def value(ls):
        return sum(int(v) << (28 * i) for i, v in enumerate(ls))

    cases = [(0, 0), (1, 1), (P - 1, P - 1), (RP % P, RP % P), (0, P - 1), ((1 << 384) - 1, (1 << 384) - 1)]
    cases += [(r.randrange(P), r.randrange(P)) for _ in range(200)]
    cases += [(r.randrange(8 * P), r.randrange(8 * P)) for _ in range(100)]       # lazy VALUES (a few p), as the point formulas keep them
    for n, (a, b) in enumerate(cases):
        la, lb = limbs(a, lazy=n % 5), limbs(b, lazy=(n // 5) % 7)
        assert value(la) == a and value(lb) == b
        out = (ctypes.c_uint32 * 16)()
        lib.h_coop_mul((ctypes.c_uint32 * 14)(*la), (ctypes.c_uint32 * 14)(*lb), out)
        got = list(out)
        assert got[14] == 0 and got[15] == 0 and all(v < (1 << 30) for v in got)
        assert value(got[:14]) % P == a * b * rinv % P, (n, a, b)
        assert value(got[:14]) < a * b // RP + 5 * P
    # the largest operands of the contract (values under 2^388, every limb but the top at the lazy maximum 2^30 - 1): the
    # column accumulators must not wrap and nothing may leave limb 13
    top = [(1 << 30) - 1] * 13 + [(1 << 22) - 1]
    assert value(top) < 1 << 388
    out = (ctypes.c_uint32 * 16)()
    lib.h_coop_mul((ctypes.c_uint32 * 14)(*top), (ctypes.c_uint32 * 14)(*top), out)
    got = list(out)
    assert got[14] == 0 and got[15] == 0 and all(v < (1 << 30) for v in got)
    assert value(got[:14]) % P == value(top) * value(top) * rinv % P


def test_partial_sums_of_a_sharded_commitment_on_host(lib):
    """hostg1.hpp h1_sum_strided: the per-rank partial sums of a commitment added in 64-bit-limb arithmetic (identity
    contributions, equal contributions -> the doubling branch, opposite ones -> the identity) against the oracle."""
    rnd = random.Random(65)
    G = E.G1_GEN
    for trial in range(8):
        n = [1, 2, 3, 4, 8, 8, 5, 2][trial]
        pts = [E.g1_mul(G, rnd.randrange(1, Q)) for _ in range(n)]
        if trial == 4:
            pts = [pts[0]] * n                     # loop-back: every rank contributes the same point
        if trial == 5:
            pts[2] = None
            pts[7] = None                          # ranks with an empty slice
        if trial == 7:
            pts[1] = E.g1_mul(pts[0], Q - 1)       # P + (-P)
        raw = b"".join(E.g1_to_raw96(p) if p is not None else bytes(96) for p in pts)
        out = (ctypes.c_uint8 * 48)()
        lib.h_sum_strided(raw, n, out)
        want = None
        for p in pts:
            if p is not None:
                want = E.g1_add(want, p)
        assert bytes(out) == E.g1_compress(want), trial


@pytest.mark.parametrize("workers", [0, 1, 3, 7])
def test_host_helper_threads_run_every_task_exactly_once(lib, workers):
    """finish_pool.hpp: jobs of 0..16 tasks under every arming pattern of fetch_commitments (armed and run, armed and
    withdrawn, not armed, workers already spinning): each task runs once, run() returns after the last."""
    assert lib.h_finish_pool_selftest(workers, 3000) == 0


def test_first_two_entries_of_a_lane_through_the_affine_pair_formula(lib):
    """curve28.cuh add_affine_pair (msm.hip ACC_FIRST_PAIR): the sum of two affine table points with 4 products + 2 squarings,
    every sign combination (signs applied lazily as 4p - y), followed by a chain of mixed additions (the output bounds must
    be add_affine's input bounds); equal and opposite first points fall back to the general path."""
    rnd = random.Random(6201)
    G = E.G1_GEN
    pts = [E.g1_mul(G, rnd.randrange(1, Q)) for _ in range(24)]
    out = (ctypes.c_uint8 * 96)()
    used = ctypes.c_int(0)
    for trial in range(40):
        n = [2, 2, 2, 2, 3, 8, 24][trial % 7]
        sel = [pts[rnd.randrange(len(pts))] for _ in range(n)]
        if trial % 7 == 0:
            sel[1] = pts[(pts.index(sel[0]) + 1) % len(pts)]
        neg = bytes([(trial >> 0) & 1, (trial >> 1) & 1] + [rnd.randrange(2) for _ in range(n - 2)])
        raw = b"".join(E.g1_to_raw96(p) for p in sel)
        ok = lib.h_g1r_accumulate_pair_first(raw, neg, n, out, ctypes.byref(used))
        exp = None
        for p, s in zip(sel, neg):
            exp = E.g1_add(exp, (p[0], (E.P - p[1]) % E.P) if s else p)
        assert (ok == 1) == (exp is not None), trial
        if exp is not None:
            assert E.g1_from_raw96(bytes(out)) == exp, trial
        assert used.value == (1 if sel[0][0] != sel[1][0] else 0), trial
    # equal first points (doubling) and opposite ones (cancellation) never take the pair formula
    rep = E.g1_to_raw96(pts[0]) * 5
    assert lib.h_g1r_accumulate_pair_first(rep, bytes(5), 5, out, ctypes.byref(used)) == 1 and used.value == 0
    assert E.g1_from_raw96(bytes(out)) == E.g1_mul(pts[0], 5)
    assert lib.h_g1r_accumulate_pair_first(rep, bytes([0, 1, 0, 0, 0]), 5, out, ctypes.byref(used)) == 1 and used.value == 0
    assert E.g1_from_raw96(bytes(out)) == E.g1_mul(pts[0], 3)


def test_scalar_multiplication_through_the_endomorphism(lib):
    """curve28.cuh glv_split / g1r_mul_glv (the scalar multiplication of the Lagrange-basis key's group FFT, msm.hip):
    k = k1 + k2 LAMBDA with both halves below 2^128 and r = LAMBDA^2 + LAMBDA + 1; phi(x, y) = (BETA x, y) = [LAMBDA](x, y);
    [k] P equals the oracle's double-and-add for random, small, extreme and half-empty scalars, also for an operand that
    comes out of additions (the butterflies' bounds) and for the identity's neighbours (k = 0, 1, r - 1)."""
    lam = 0xac45a4010001a40200000000ffffffff
    assert lam * lam + lam + 1 == Q
    beta = 0x1a0111ea397fe699ec02408663d4de85aa0d857d89759ad4897d29650fb85f9b409427eb4f49fffd8bfd00000000aaac
    G = E.G1_GEN
    assert pow(beta, 3, E.P) == 1 and (beta * G[0] % E.P, G[1]) == E.g1_mul(G, lam)
    rnd = random.Random(6202)
    ks = [0, 1, 2, lam - 1, lam, lam + 1, 2 * lam, lam * lam, Q - 1, Q - lam, (1 << 128) - 1, 1 << 128, (1 << 254) + 1]
    ks += [rnd.randrange(Q) for _ in range(40)] + [rnd.randrange(1 << 64) for _ in range(4)] + [rnd.randrange(1 << 64) * lam % Q for _ in range(4)]
    out4 = (ctypes.c_uint64 * 4)()
    out = (ctypes.c_uint8 * 96)()
    for i, k in enumerate(ks):
        limbs = (ctypes.c_uint32 * 8)(*[(k >> (32 * j)) & 0xffffffff for j in range(8)])
        lib.h_glv_split(limbs, out4)
        k1, k2 = out4[0] | out4[1] << 64, out4[2] | out4[3] << 64
        assert (k1, k2) == (k % lam, k // lam), hex(k)
        base = E.g1_mul(G, rnd.randrange(1, Q))
        pre = i % 3
        ok = lib.h_g1r_mul_glv(E.g1_to_raw96(base), limbs, pre, out)
        exp = E.g1_mul(base, k * (1 << pre) % Q) if k else None
        assert (ok == 1) == (exp is not None), hex(k)
        if exp is not None:
            assert E.g1_from_raw96(bytes(out)) == exp, hex(k)
