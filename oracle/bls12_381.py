"""BLS12-381 scalar field Fr, base field Fp and G1 — big-int restatement (oracle).

Replaces (for checking only) the external crate `dusk-bls12_381 0.14`
(reference Cargo.toml:23), which supplies BlsScalar, G1Affine/G1Projective and
`multiscalar_mul::msm_variable_base` used at reference
src/commitment_scheme/kzg10/key.rs:14,384 and src/fft/domain.rs:17,115.

Conventions pinned by the reference tree itself:
  * BlsScalar.0 = 4 x u64 little-endian limbs in Montgomery form, R = 2^256:
    the MINUS_ONE literal at src/composer.rs:334-339 equals (-R mod q) limbs
    (checked in tests/test_oracle_fields.py).
  * from_raw([7,0,0,0]) = canonical 7 (src/composer/permutation/constants.rs:14).
"""
from __future__ import annotations

# ----------------------------------------------------------------------------
# Scalar field Fr
# ----------------------------------------------------------------------------
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
FR_R = (1 << 256) % Q
FR_R2 = (FR_R * FR_R) % Q
FR_RINV = pow(FR_R, -1, Q)
TWO_ADACITY = 32                       # domain.rs:115 (imported constant)
GENERATOR = 7                          # multiplicative generator, domain.rs:115
ROOT_OF_UNITY = pow(GENERATOR, (Q - 1) >> TWO_ADACITY, Q)
K1, K2, K3 = 7, 13, 17                 # permutation/constants.rs:14-16
# JubJub twisted-Edwards d = -(10240/10241) (dusk_jubjub::EDWARDS_D, used at
# widget/ecc/**/proverkey.rs).  Upstream constant; not exercised by the KAT.
EDWARDS_D = (-10240 * pow(10241, -1, Q)) % Q


def fr_inv(a: int) -> int:
    return pow(a, -1, Q)


def fr_to_mont_limbs(a: int) -> list[int]:
    """Canonical int -> BlsScalar.0 (4 x u64 LE Montgomery limbs)."""
    m = (a * FR_R) % Q
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def fr_from_mont_limbs(l) -> int:
    m = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return (m * FR_RINV) % Q


def fr_to_bytes(a: int) -> bytes:
    """BlsScalar::to_bytes — canonical 32-byte little-endian."""
    return int(a % Q).to_bytes(32, "little")


def fr_from_bytes_wide(b: bytes) -> int:
    """BlsScalar::from_bytes_wide — 512-bit LE integer reduced mod q
    (used by transcript.rs:98-103 and BlsScalar::random, util.rs:135-161)."""
    assert len(b) == 64
    return int.from_bytes(b, "little") % Q


# ----------------------------------------------------------------------------
# Base field Fp and G1:  y^2 = x^3 + 4
# ----------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FP_R = (1 << 384) % P
FP_RINV = pow(FP_R, -1, P)
G1_B = 4
G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
assert (G1_GEN[1] ** 2 - G1_GEN[0] ** 3 - G1_B) % P == 0


def fp_to_mont_limbs(a: int) -> list[int]:
    m = (a * FP_R) % P
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def fp_from_mont_limbs(l) -> int:
    m = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return (m * FP_RINV) % P


# Points: affine = (x, y) or None for identity; jacobian = (X, Y, Z), Z == 0 identity.
JAC_ID = (1, 1, 0)


def to_jac(pt):
    return JAC_ID if pt is None else (pt[0], pt[1], 1)


def jac_double(p):
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return JAC_ID
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def jac_add(p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == 0:
        return q
    if Z2 == 0:
        return p
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return jac_double(p)
        return JAC_ID
    H = (U2 - U1) % P
    I = 4 * H * H % P
    J = H * I % P
    r = 2 * (S2 - S1) % P
    V = U1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * S1 * J) % P
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P
    return (X3, Y3, Z3)


def jac_neg(p):
    return (p[0], (-p[1]) % P, p[2])


def jac_mul(p, k: int):
    k %= Q
    acc = JAC_ID
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, p)
    return acc


def to_affine(p):
    X, Y, Z = p
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def batch_to_affine(ps):
    """G1Projective::batch_normalize (srs.rs:87-88) — mathematically per-point."""
    return [to_affine(p) for p in ps]


def g1_mul(pt, k: int):
    return to_affine(jac_mul(to_jac(pt), k))


def g1_add(a, b):
    return to_affine(jac_add(to_jac(a), to_jac(b)))


def g1_compress(pt) -> bytes:
    """G1Affine::to_bytes (commitment.rs:49-51): 48-byte big-endian x with flag
    bits 0x80 compressed | 0x40 infinity | 0x20 y lexicographically largest."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P - y) % P:
        b[0] |= 0x20
    return bytes(b)


def g1_decompress(b: bytes):
    assert len(b) == 48 and b[0] & 0x80
    if b[0] & 0x40:
        return None
    sign = bool(b[0] & 0x20)
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    y2 = (x * x * x + G1_B) % P
    y = pow(y2, (P + 1) // 4, P)
    assert y * y % P == y2, "not on curve"
    if (y > (P - y) % P) != sign:
        y = (P - y) % P
    return (x, y)


def g1_to_raw96(pt) -> bytes:
    """x||y as 2 x 6 x u64 LE Montgomery limbs — the first 96 bytes of
    G1Affine::to_raw_bytes (key.rs:215-229: 97 B = x, y, infinity flag)."""
    assert pt is not None
    out = b""
    for c in pt:
        for l in fp_to_mont_limbs(c):
            out += l.to_bytes(8, "little")
    return out


def g1_from_raw96(b: bytes):
    xs = [int.from_bytes(b[8 * i:8 * i + 8], "little") for i in range(12)]
    return (fp_from_mont_limbs(xs[:6]), fp_from_mont_limbs(xs[6:]))


def msm_naive(points, scalars):
    """Definition of msm_variable_base(points, scalars) (key.rs:384): the
    result is a unique group element, independent of the algorithm.  The
    dependency zips the slices, so extra bases are ignored."""
    acc = JAC_ID
    for pt, s in zip(points, scalars):
        if s % Q and pt is not None:
            acc = jac_add(acc, jac_mul(to_jac(pt), s))
    return to_affine(acc)


def msm_pippenger(points, scalars, c: int | None = None):
    """Textbook Pippenger (bucket method) — same result as msm_naive, used for
    larger oracle sizes.  Window choice mirrors the arkworks/zexe heuristic the
    dependency ships, but the result does not depend on it."""
    m = min(len(points), len(scalars))
    if m == 0:
        return None
    if c is None:
        c = 3 if m < 32 else max(3, (m.bit_length() - 1) * 69 // 100 + 2)
    nwin = (255 + c - 1) // c
    total = JAC_ID
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = jac_double(total)
        buckets = [JAC_ID] * ((1 << c) - 1)
        for pt, s in zip(points[:m], scalars[:m]):
            d = ((s % Q) >> (w * c)) & ((1 << c) - 1)
            if d and pt is not None:
                buckets[d - 1] = jac_add(buckets[d - 1], to_jac(pt))
        run = JAC_ID
        acc = JAC_ID
        for b in reversed(buckets):
            run = jac_add(run, b)
            acc = jac_add(acc, run)
        total = jac_add(total, acc)
    return to_affine(total)
