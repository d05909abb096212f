// G1 in XYZZ coordinates over the reduced-radix field Fp28 (fp28.cuh), lazily reduced.
//
// Same group law as curve.cuh (EFD xyzz madd-2008-s / add-2008-s / dbl-2008-s-1); the
// difference is bookkeeping: no conditional subtractions — every coordinate carries a static
// bound (in multiples of p) that is closed under the three operations:
//        X < 16p,  Y < 8p,  ZZ < 2p,  ZZZ < 2p        (table points: x, y < 2p)
// Subtractions add the smallest pad K*p with K >= 2 * bound(subtrahend); every product's
// operand bounds multiply to < 2528 (= R'/p), so products come out < 2p.  The bounds are
// annotated on each line.  ZZ == 0 (all limbs) marks the identity.
#pragma once
#include "curve.cuh"
#include "fp28.cuh"

namespace plonk {

struct alignas(16) Fp28Slot {   // 14 limbs + 2 pad words = 64 B so coordinates stay 16-B aligned
  uint32_t w[16];
};
struct alignas(16) G1AffineR {  // 128 B table entry: one cache line per gather
  Fp28Slot x, y;
};

struct G1R {
  Fp28 X, Y, ZZ, ZZZ;

  HD static G1R identity() {
    G1R r;
    r.X = Fp28::one();
    r.Y = Fp28::one();
    r.ZZ = Fp28::zero();
    r.ZZZ = Fp28::zero();
    return r;
  }
  HD bool is_identity() const { return ZZ.is_zero_limbs(); }

  HD static G1R from_affine(const Fp28& x, const Fp28& y) {
    G1R r;
    r.X = x;
    r.Y = y;
    r.ZZ = Fp28::one();
    r.ZZZ = Fp28::one();
    return r;
  }

  // cheap necessary condition for v == 0 (mod p) when value(v) < 64p: v = k p with k < 64,
  // so (v mod 2^28) * p^-1 mod 2^28 = k < 64.  False positives 2^-22 -> exact test.
  HD static bool maybe_zero(const Fp28& v) {
    const uint32_t pinv = (0u - Fp28::INV) & Fp28::MASK;   // p^-1 mod 2^28
    return ((v.l[0] * pinv) & Fp28::MASK) < 64u;
  }
  HD static bool zero_mod(const Fp28& v) { return maybe_zero(v) && v.is_zero_mod(); }

  // 2 * this.  In: X<16p Y<8p ZZ,ZZZ<2p.  Out: X<10p Y<2p ZZ,ZZZ<2p.
  HD G1R dbl() const {
    if (is_identity()) return identity();   // (Y == 0 cannot happen in the prime-order group)
    const Fp28 U = Y.dbl();                                   // < 16p
    const Fp28 V = U.sqr();                                   // 16*16            -> < 2p
    const Fp28 W = Fp28::mul(U, V);                           // 16*2             -> < 2p
    const Fp28 S = Fp28::mul(X, V);                           // 16*2             -> < 2p
    const Fp28 XX = X.sqr();                                  // 16*16            -> < 2p
    const Fp28 M = Fp28::add(XX.dbl(), XX);                   // < 6p
    G1R r;
    r.X = Fp28::sub<8>(M.sqr(), Fp28::add_lazy(S, S));        // 2p - (<4p) + 8p   -> < 10p
    r.Y = Fp28::mul2(M, Fp28::sub_lazy<32>(S, r.X),           // 6 * (2+32=34) = 204
                     W, Fp28::neg_lazy<16>(Y));               // + 2 * 16 = 236    -> < 2p
    r.ZZ = Fp28::mul(V, ZZ);                                  // < 2p
    r.ZZZ = Fp28::mul(W, ZZZ);                                // < 2p
    return r;
  }
  HD static G1R dbl_affine(const Fp28& x, const Fp28& y) { return from_affine(x, y).dbl(); }

  // this + (x2, y2), affine operand never the identity; x2 < 2p, y2 < 4p (after negation; y2
  // may carry lazy limbs, it is normalised only on the rare paths that store it).
  // 8 products + 2 squarings, 9 Montgomery reductions (Y3 is one fused two-product reduction).
  // In: X<16p Y<8p.  Out: X<14p Y<2p.
  HD G1R add_affine(const Fp28& x2, const Fp28& y2) const {
    if (is_identity()) return from_affine(x2, y2.normalized());
    const Fp28 U2 = Fp28::mul(x2, ZZ);                        // 2*2               -> < 2p
    const Fp28 S2 = Fp28::mul(y2, ZZZ);                       // 4*2               -> < 2p
    const Fp28 P_ = Fp28::sub_lazy<32>(U2, X);                // X<16p  -> < 34p, lazy limbs
    const Fp28 R_ = Fp28::sub<16>(S2, Y);                     // Y<8p   -> < 18p, normalised
    if (maybe_zero(P_) && P_.normalized().is_zero_mod()) {
      if (R_.is_zero_mod()) return dbl_affine(x2, y2.normalized());
      return identity();
    }
    const Fp28 PP = P_.sqr();                                 // 34*34 = 1156      -> < 2p
    const Fp28 PPP = Fp28::mul(P_, PP);                       // 34*2              -> < 2p
    const Fp28 Q_ = Fp28::mul(X, PP);                         // 16*2              -> < 2p
    G1R r;
    r.X = Fp28::sub<8>(Fp28::sub_lazy<4>(R_.sqr(), PPP),      // 18*18=324; 2p + 4p
                       Fp28::add_lazy(Q_, Q_));               // - (<4p) + 8p      -> < 14p
    r.Y = Fp28::mul2(R_, Fp28::sub_lazy<32>(Q_, r.X),         // 18 * (2+32=34) = 612
                     PPP, Fp28::neg_lazy<16>(Y));             // + 2 * 16 = 644    -> < 2p
    r.ZZ = Fp28::mul(ZZ, PP);                                 // < 2p
    r.ZZZ = Fp28::mul(ZZZ, PPP);                              // < 2p
    return r;
  }

  // (x1, y1) + (x2, y2), BOTH affine and neither the identity: the first addition of every accumulation lane.  With
  // ZZ = ZZZ = 1 on the left, four of add_affine's ten products are multiplications by one (U2 = x2, S2 = y2, ZZ3 = PP,
  // ZZZ3 = PPP): 4 products + 2 squarings, 5 Montgomery reductions.  x1, x2 < 2p normalised; y1, y2 < 4p, possibly lazy
  // (the sign of a table entry is applied as 4p - y without carry propagation).  `pair_distinct` must hold — equal or
  // opposite points (x1 = x2) take the general path, which doubles or cancels.  Out: X<14p Y<2p ZZ,ZZZ<2p, as add_affine.
  HD static bool pair_distinct(const Fp28& x1, const Fp28& x2) { return !maybe_zero(Fp28::sub_lazy<4>(x2, x1)); }   // (no false "distinct": exact zero always passes maybe_zero)
  HD static G1R add_affine_pair(const Fp28& x1, const Fp28& y1, const Fp28& x2, const Fp28& y2) {
    const Fp28 P_ = Fp28::sub_lazy<4>(x2, x1);                // < 6p, lazy limbs
    const Fp28 y1n = y1.normalized();                         // < 4p
    const Fp28 R_ = Fp28::sub<8>(y2, y1n);                    // 4p - (<4p) + 8p   -> < 12p, normalised
    const Fp28 PP = P_.sqr();                                 // 6*6               -> < 2p
    const Fp28 PPP = Fp28::mul(P_, PP);                       // 6*2               -> < 2p
    const Fp28 Q_ = Fp28::mul(x1, PP);                        // 2*2               -> < 2p
    G1R r;
    r.X = Fp28::sub<8>(Fp28::sub_lazy<4>(R_.sqr(), PPP),      // 12*12 = 144; 2p + 4p
                       Fp28::add_lazy(Q_, Q_));               // - (<4p) + 8p      -> < 14p
    r.Y = Fp28::mul2(R_, Fp28::sub_lazy<32>(Q_, r.X),         // 12 * (2+32=34) = 408
                     PPP, Fp28::neg_lazy<8>(y1n));            // + 2 * 8 = 424     -> < 2p
    r.ZZ = PP;
    r.ZZZ = PPP;
    return r;
  }

  // full addition (same bounds in and out)
  HD G1R add(const G1R& b) const {
    if (is_identity()) return b;
    if (b.is_identity()) return *this;
    const Fp28 U1 = Fp28::mul(X, b.ZZ);                       // 16*2 -> < 2p
    const Fp28 U2 = Fp28::mul(b.X, ZZ);
    const Fp28 S1 = Fp28::mul(Y, b.ZZZ);                      // 8*2  -> < 2p
    const Fp28 S2 = Fp28::mul(b.Y, ZZZ);
    const Fp28 P_ = Fp28::sub_lazy<4>(U2, U1);                // < 6p, lazy limbs
    const Fp28 R_ = Fp28::sub<4>(S2, S1);                     // < 6p, normalised
    if (maybe_zero(P_) && P_.normalized().is_zero_mod()) {
      if (R_.is_zero_mod()) return dbl();
      return identity();
    }
    const Fp28 PP = P_.sqr();
    const Fp28 PPP = Fp28::mul(P_, PP);
    const Fp28 Q_ = Fp28::mul(U1, PP);
    G1R r;
    r.X = Fp28::sub<8>(Fp28::sub_lazy<4>(R_.sqr(), PPP), Fp28::add_lazy(Q_, Q_));   // < 14p
    r.Y = Fp28::mul2(R_, Fp28::sub_lazy<32>(Q_, r.X), PPP, Fp28::neg_lazy<4>(S1));  // 6*34 + 2*4 -> < 2p
    r.ZZ = Fp28::mul(Fp28::mul(ZZ, b.ZZ), PP);
    r.ZZZ = Fp28::mul(Fp28::mul(ZZZ, b.ZZZ), PPP);
    return r;
  }

  HD G1R mul_u32(uint32_t k) const {
    G1R acc = identity();
    bool started = false;
    for (int b = 31; b >= 0; --b) {
      if (started) acc = acc.dbl();
      if ((k >> b) & 1) {
        acc = started ? acc.add(*this) : *this;
        started = true;
      }
    }
    return acc;
  }

  // back to the 12 x 32-bit form (canonical coordinates) for the host / ABI
  HD G1 to_g1() const {
    G1 r;
    if (is_identity()) return G1::identity();
    r.X = X.to_fp();
    r.Y = Y.to_fp();
    r.ZZ = ZZ.to_fp();
    r.ZZZ = ZZZ.to_fp();
    return r;
  }
};

// Fermat inverse in Fp28 (value < 2p in, < 2p out)
HD Fp28 fp28_inv(const Fp28& a) {
  // exponent p - 2 as 32-bit words
  uint32_t e[12];
  uint64_t borrow = 2;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    uint64_t t = (uint64_t)FpP::MOD[i] - borrow;
    e[i] = (uint32_t)t;
    borrow = (t >> 63) & 1;
  }
  Fp28 acc = Fp28::one();
  bool started = false;
  for (int w = 11; w >= 0; --w)
    for (int b = 31; b >= 0; --b) {
      if (started) acc = acc.sqr();
      if ((e[w] >> b) & 1) {
        acc = started ? Fp28::mul(acc, a) : a;
        started = true;
      }
    }
  return acc;
}

}  // namespace plonk
#include "fp_safegcd.cuh"
namespace plonk {

// ---- [k] P for a full-width scalar through the curve's endomorphism (setup only: the group FFT of the Lagrange-basis key) ----
// phi(x, y) = (BETA x, y) = [LAMBDA] (x, y) on G1, LAMBDA = z^2 - 1 for the curve parameter z = -0xd201000000010000, and the
// group order is r = LAMBDA^2 + LAMBDA + 1 EXACTLY, so k = k1 + k2 LAMBDA with k2 = floor(k / LAMBDA) <= LAMBDA + 1 < 2^128 and
// k1 = k mod LAMBDA < 2^128: [k] P = [k1] P + [k2] phi(P) is one interleaved double-and-add of 128 steps over {P, phi(P),
// P + phi(P)} — 128 doublings + ~96 additions instead of 255 + ~127.
struct GlvScalar { uint64_t k1[2], k2[2]; };
// k (8 x 32-bit limbs, canonical, < r) -> (k mod LAMBDA, k div LAMBDA): restoring division, 255 steps on a 129-bit remainder
HD GlvScalar glv_split(const uint32_t* k /*8*/) {
  constexpr uint64_t LH = 0xac45a4010001a402ull, LL = 0x00000000ffffffffull;   // LAMBDA
  uint64_t rh = 0, rl = 0, qh = 0, ql = 0;
  for (int i = 254; i >= 0; --i) {
    const uint64_t top = rh >> 63;                            // the remainder is < LAMBDA < 2^128 before the shift, < 2^129 after
    rh = (rh << 1) | (rl >> 63);
    rl = (rl << 1) | ((k[i >> 5] >> (i & 31)) & 1u);
    const bool ge = top || rh > LH || (rh == LH && rl >= LL);
    if (ge) {                                                 // < 2 LAMBDA - LAMBDA: fits 128 bits again
      const uint64_t borrow = rl < LL ? 1u : 0u;
      rl -= LL;
      rh -= LH + borrow;
    }
    qh = (qh << 1) | (ql >> 63);                              // (the quotient's bits above 127 are zero: k < r)
    ql = (ql << 1) | (ge ? 1u : 0u);
  }
  GlvScalar g;
  g.k1[0] = rl; g.k1[1] = rh; g.k2[0] = ql; g.k2[1] = qh;
  return g;
}
HD Fp28 glv_beta() {   // BETA * R' mod p, BETA = 0x1a0111ea...aaac (the cube root of unity that goes with LAMBDA: oracle check in tests/test_field_host.py)
  constexpr uint32_t V[Fp28::N] = {0x2421b59u, 0xbee4867u, 0x1d31002u, 0x4760184u, 0x4cc5086u, 0xc76dc00u, 0xaae891bu,
                                   0xac70ad2u, 0xfe377c4u, 0xe4686b8u, 0x5ed1568u, 0x8f5a180u, 0x02b5c1fu, 0x000d1a4u};
  Fp28 r;
#pragma unroll
  for (int i = 0; i < Fp28::N; ++i) r.l[i] = V[i];
  return r;
}
// [k] p, k canonical (8 x 32-bit limbs, < r); p with the bounds of G1R::add's operands
HD G1R g1r_mul_glv(const G1R& p, const uint32_t* k /*8*/) {
  const GlvScalar g = glv_split(k);
  G1R p2 = p;
  p2.X = Fp28::mul(p.X, glv_beta());                          // 16 * 1 -> < 2p
  const G1R p3 = p.add(p2);
  G1R acc = G1R::identity();
  for (int b = 127; b >= 0; --b) {
    acc = acc.dbl();
    const uint32_t s = (uint32_t)((g.k1[b >> 6] >> (b & 63)) & 1u) | ((uint32_t)((g.k2[b >> 6] >> (b & 63)) & 1u) << 1);
    if (s) acc = acc.add(s == 1 ? p : (s == 2 ? p2 : p3));
  }
  return acc;
}

// affine coordinates (x, y) < 2p of a finite point (safegcd inverse: ~23 k instructions instead of ~300 k)
HD void g1r_to_affine(const G1R& p, Fp28* x, Fp28* y) {
  const Fp28 inv = fp28_inv_gcd(Fp28::mul(p.ZZ, p.ZZZ));
  const Fp28 izz = Fp28::mul(inv, p.ZZZ);
  const Fp28 izzz = Fp28::mul(inv, p.ZZ);
  *x = Fp28::mul(p.X, izz);
  *y = Fp28::mul(p.Y, izzz);
}

}  // namespace plonk
