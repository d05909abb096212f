// Host-side validity test of a compressed G2 encoding: what OpeningKey::from_bytes (reference src/commitment_scheme/
// kzg10/key.rs:596-615) asks of `h` and `x_h` through G2Affine::from_bytes of the dusk-bls12_381 dependency (zkcrypto
// layout: 96 bytes, x.c1 then x.c0 big-endian, flag bits 0x80 compressed / 0x40 infinity / 0x20 sign of y in the first
// byte).  An encoding is accepted iff
//   * the compression flag is set;
//   * infinity flag set:  the sign flag is clear and every other bit is zero  (the identity decodes HERE; OpeningKey::try_new,
//     key.rs:617-648, then refuses an identity g, h or x_h — serial.hip public_parameters_check does that before calling this);
//   * otherwise: both coordinates of x are canonical (< p), x^3 + 4 (1 + u) is a square in Fp2 (either root is a point;
//     the sign flag only picks one of them), and the point has order r  ([r] P = O, is_torsion_free).
// No G2 arithmetic exists anywhere else in the library (the prover never touches the opening key): this runs once per
// loaded parameter file, on the host, with the 64-bit-limb Fp of hostg1.hpp (~0.3 ms per point).
#pragma once
#include "hostg1.hpp"

namespace plonk {

struct F2 {   // a + b u,  u^2 = -1
  Fp64 a, b;
};
static inline F2 f2_add(const F2& x, const F2& y) { return {fp64_add(x.a, y.a), fp64_add(x.b, y.b)}; }
static inline F2 f2_sub(const F2& x, const F2& y) { return {fp64_sub(x.a, y.a), fp64_sub(x.b, y.b)}; }
static inline F2 f2_dbl(const F2& x) { return f2_add(x, x); }
static inline F2 f2_mul(const F2& x, const F2& y) {
  const Fp64 aa = fp64_mul(x.a, y.a), bb = fp64_mul(x.b, y.b);
  const Fp64 cross = fp64_mul(fp64_add(x.a, x.b), fp64_add(y.a, y.b));   // aa + bb + (a b' + a' b)
  return {fp64_sub(aa, bb), fp64_sub(fp64_sub(cross, aa), bb)};
}
static inline F2 f2_sqr(const F2& x) {
  const Fp64 ab = fp64_mul(x.a, x.b);
  return {fp64_mul(fp64_add(x.a, x.b), fp64_sub(x.a, x.b)), fp64_add(ab, ab)};
}
static inline bool f2_is_zero(const F2& x) { return fp64_is_zero(x.a) && fp64_is_zero(x.b); }
static inline bool f2_eq(const F2& x, const F2& y) { return f2_is_zero(f2_sub(x, y)); }
static F2 f2_one() {
  Fp64 z;
  memset(&z, 0, sizeof z);
  return {to64(Fp::one()), z};
}
static F2 f2_pow(const F2& x, const Fp64& e) {
  F2 acc = f2_one();
  for (int w = 5; w >= 0; --w)
    for (int b = 63; b >= 0; --b) {
      acc = f2_sqr(acc);
      if ((e.l[w] >> b) & 1) acc = f2_mul(acc, x);
    }
  return acc;
}
// square root in Fp2 for p = 3 mod 4 (Adj, Rodriguez-Henriquez, "Square root computation over even extension fields",
// algorithm 9): false when v is not a square
static bool f2_sqrt(const F2& v, F2* out) {
  const Fp64 M = fp64_mod();
  Fp64 e1 = M, e2 = M;                       // (p - 3) / 4 and (p - 1) / 2; p = ...aaab: no borrow out of limb 0
  e1.l[0] -= 3;
  for (int i = 0; i < 6; ++i) e1.l[i] = (e1.l[i] >> 2) | (i < 5 ? e1.l[i + 1] << 62 : 0);
  e2.l[0] -= 1;
  for (int i = 0; i < 6; ++i) e2.l[i] = (e2.l[i] >> 1) | (i < 5 ? e2.l[i + 1] << 63 : 0);
  const F2 a1 = f2_pow(v, e1);
  const F2 alpha = f2_mul(f2_sqr(a1), v);
  const F2 x0 = f2_mul(a1, v);
  Fp64 zero;
  memset(&zero, 0, sizeof zero);
  F2 x;
  if (f2_is_zero(f2_add(alpha, f2_one()))) {  // alpha == -1:  x = u * x0
    x = {fp64_sub(zero, x0.b), x0.a};
  } else {
    x = f2_mul(f2_pow(f2_add(alpha, f2_one()), e2), x0);
  }
  *out = x;
  return f2_eq(f2_sqr(x), v);
}

// Jacobian coordinates over Fp2, a = 0 (EFD dbl-2009-l / add-2007-bl); Z == 0 marks the identity
struct H2 {
  F2 X, Y, Z;
  bool inf() const { return f2_is_zero(Z); }
};
static H2 h2_dbl(const H2& p) {
  if (p.inf()) return p;
  const F2 A = f2_sqr(p.X), B = f2_sqr(p.Y), C = f2_sqr(B);
  const F2 D = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(p.X, B)), A), C));
  const F2 E = f2_add(f2_dbl(A), A), F = f2_sqr(E);
  H2 r;
  r.X = f2_sub(F, f2_dbl(D));
  r.Y = f2_sub(f2_mul(E, f2_sub(D, r.X)), f2_dbl(f2_dbl(f2_dbl(C))));
  r.Z = f2_dbl(f2_mul(p.Y, p.Z));
  return r;
}
static H2 h2_add(const H2& p, const H2& q) {
  if (p.inf()) return q;
  if (q.inf()) return p;
  const F2 Z1Z1 = f2_sqr(p.Z), Z2Z2 = f2_sqr(q.Z);
  const F2 U1 = f2_mul(p.X, Z2Z2), U2 = f2_mul(q.X, Z1Z1);
  const F2 S1 = f2_mul(f2_mul(p.Y, q.Z), Z2Z2), S2 = f2_mul(f2_mul(q.Y, p.Z), Z1Z1);
  if (f2_eq(U1, U2)) {
    if (f2_eq(S1, S2)) return h2_dbl(p);
    H2 o = p;
    memset(&o.Z, 0, sizeof o.Z);
    return o;
  }
  const F2 H = f2_sub(U2, U1), I = f2_sqr(f2_dbl(H)), J = f2_mul(H, I);
  const F2 R = f2_dbl(f2_sub(S2, S1)), V = f2_mul(U1, I);
  H2 r;
  r.X = f2_sub(f2_sub(f2_sqr(R), J), f2_dbl(V));
  r.Y = f2_sub(f2_mul(R, f2_sub(V, r.X)), f2_dbl(f2_mul(S1, J)));
  r.Z = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
  return r;
}

// 48 big-endian bytes (the top three bits of the first byte masked off when `mask_flags`) -> raw limbs; false if >= p
static bool fp64_from_be48(const uint8_t in[48], bool mask_flags, Fp64* out) {
  Fp64 raw;
  for (int i = 0; i < 6; ++i) {
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b) {
      const int pos = 8 * (5 - i) + b;
      v = (v << 8) | ((pos == 0 && mask_flags) ? (uint8_t)(in[0] & 0x1f) : in[pos]);
    }
    raw.l[i] = v;
  }
  const Fp64 M = fp64_mod();
  bool lt = false;
  for (int i = 5; i >= 0; --i)
    if (raw.l[i] != M.l[i]) { lt = raw.l[i] < M.l[i]; break; }
  *out = raw;
  return lt;
}

static bool g2_compressed_valid(const uint8_t in[96]) {
  const uint8_t flags = in[0];
  if (!(flags & 0x80)) return false;
  Fp64 c1, c0;
  const bool canon1 = fp64_from_be48(in, true, &c1), canon0 = fp64_from_be48(in + 48, false, &c0);
  if (flags & 0x40) return !(flags & 0x20) && fp64_is_zero(c1) && fp64_is_zero(c0);
  if (!canon1 || !canon0) return false;
  Fp r2;
  for (int i = 0; i < 12; ++i) r2.l[i] = FpP::R2[i];
  const F2 x = {fp64_mul(c0, to64(r2)), fp64_mul(c1, to64(r2))};          // Montgomery form
  const Fp64 four = to64(Fp::from_u64(4));
  const F2 rhs = f2_add(f2_mul(f2_sqr(x), x), F2{four, four});              // x^3 + 4 (1 + u)
  F2 y;
  if (!f2_sqrt(rhs, &y)) return false;                                      // not on the twist curve
  H2 P, acc;
  P.X = x; P.Y = y; P.Z = f2_one();
  memset(&acc, 0, sizeof acc);
  for (int b = 254; b >= 0; --b) {                                          // [r] P == O  (is_torsion_free)
    acc = h2_dbl(acc);
    if ((FrP::MOD[b >> 5] >> (b & 31)) & 1) acc = h2_add(acc, P);
  }
  return acc.inf();
}

}  // namespace plonk
