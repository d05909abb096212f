"""Plain-Python G2 of BLS12-381 for the tests of the opening-key validation (plonk_amd/csrc/hostg2.hpp): Fp2 = Fp[u] / (u^2 + 1),
E'(Fp2): y^2 = x^3 + 4 (1 + u), affine arithmetic with modular inverses, the zkcrypto / zcash 96-byte compressed encoding
(x.c1 then x.c0 big-endian; 0x80 compressed, 0x40 infinity, 0x20 y lexicographically largest).  Test infrastructure only."""
from oracle.bls12_381 import P, Q

B2 = (4, 4)
# the standard generator of G2 (checked on the curve and of order Q by tests/test_field_host.py)
G2_GEN = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
           0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
          (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
           0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))


def f2_add(x, y): return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
def f2_sub(x, y): return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
def f2_mul(x, y): return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
def f2_sqr(x): return f2_mul(x, x)
def f2_neg(x): return ((-x[0]) % P, (-x[1]) % P)


def f2_inv(x):
    n = pow(x[0] * x[0] + x[1] * x[1], -1, P)
    return (x[0] * n % P, (-x[1]) * n % P)


def f2_pow(x, e):
    acc = (1, 0)
    for bit in bin(e)[2:]:
        acc = f2_sqr(acc)
        if bit == "1":
            acc = f2_mul(acc, x)
    return acc


def f2_sqrt(v):
    """a square root of v in Fp2, or None (p = 3 mod 4)"""
    a1 = f2_pow(v, (P - 3) // 4)
    alpha = f2_mul(f2_sqr(a1), v)
    x0 = f2_mul(a1, v)
    if alpha == (P - 1, 0):
        x = ((-x0[1]) % P, x0[0])
    else:
        x = f2_mul(f2_pow(f2_add(alpha, (1, 0)), (P - 1) // 2), x0)
    return x if f2_sqr(x) == v else None


def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sqr(y) == f2_add(f2_mul(f2_sqr(x), x), B2)


def g2_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if p[1] != q[1] or p[1] == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_sqr(p[0])), f2_inv(f2_add(p[1], p[1])))
    else:
        lam = f2_mul(f2_sub(q[1], p[1]), f2_inv(f2_sub(q[0], p[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), p[0]), q[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(p[0], x3)), p[1]))


def g2_mul(p, k):
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, p)
    return acc


def lexicographically_largest(y):
    """zkcrypto Fp2::lexicographically_largest: c1 decides, c0 when c1 == 0"""
    half = (P - 1) // 2
    return y[1] > half or (y[1] == 0 and y[0] > half)


def g2_compress(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    x, y = pt
    out = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    out[0] |= 0x80 | (0x20 if lexicographically_largest(y) else 0)
    return bytes(out)


def g2_decompress(b: bytes, check_subgroup=True):
    """G2Affine::from_compressed: the point, None for the identity; raises ValueError on an invalid encoding"""
    if len(b) != 96 or not b[0] & 0x80:
        raise ValueError("not a compressed G2 encoding")
    c1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    c0 = int.from_bytes(b[48:], "big")
    if b[0] & 0x40:
        if b[0] & 0x20 or c1 or c0:
            raise ValueError("malformed identity")
        return None
    if c1 >= P or c0 >= P:
        raise ValueError("non-canonical coordinate")
    x = (c0, c1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
    if y is None:
        raise ValueError("not on the curve")
    if lexicographically_largest(y) != bool(b[0] & 0x20):
        y = f2_neg(y)
    if check_subgroup and g2_mul((x, y), Q) is not None:
        raise ValueError("not in the subgroup")
    return (x, y)
