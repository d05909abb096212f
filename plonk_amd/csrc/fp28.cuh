// BLS12-381 Fp in reduced radix for the MI355X integer pipe: 14 limbs of 28 bits, Montgomery
// with R' = 2^392, lazily reduced.
//
// Why: on gfx950 v_mad_u64_u32 issues at the same rate as a 32-bit integer add (one
// wave-instruction per 4 cycles; tools/ubench/rates.hip).  With full 32-bit limbs every
// product needs a carry instruction next to its mad; with 28-bit limbs 28 products of < 2^56
// fit a 64-bit column accumulator, so a Montgomery product is 2 * 14 * 14 mads plus ~4
// instructions per row and nothing else.  Values are kept "lazy": limbs < 2^28 after every
// operation (carry-normalised), but the value only bounded by a small multiple of p, which
// removes every conditional subtraction from the point formulas (bounds are documented per
// function; R'/p > 2528, so a product of operands bounded by a*p and b*p with a*b < 2528 comes
// out < 2p).
//
// Interop: Fp (12 x 32-bit limbs, R = 2^384 — the reference's raw G1Affine form) converts with
// one Montgomery product each way (from_fp / to_fp).  Used only inside the MSM (msm.hip).
#pragma once
#include "field.cuh"

namespace plonk {

struct Fp28 {
  static constexpr int N = 14;
  static constexpr int B = 28;
  static constexpr uint32_t MASK = (1u << B) - 1;
  static constexpr uint32_t INV = 0xffcfffdu;   // -p^-1 mod 2^28
  uint32_t l[N];

  HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t M[N] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                               0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
    return M[i];
  }
  // K*p with every limb but the top >= 2^29 - 2 so that (a + PAD_K - b) never borrows for a
  // normalised b < (K/2) p
  template <int K>
  HD static constexpr uint32_t pad(int i) {
    constexpr uint32_t P2[N] = {0x2fff5556u, 0x2fdffffdu, 0x27ffff71u, 0x2fffd628u, 0x2c483d55u, 0x241ed61cu, 0x2ece61a3u,
                                0x2e70a255u, 0x28ee9707u, 0x29759aeau, 0x274f6c84u, 0x2cd34961u, 0x23d472fdu, 0x00034020u};
    constexpr uint32_t P4[N] = {0x2ffeaaacu, 0x2fbffffdu, 0x2ffffee5u, 0x2fffac52u, 0x28907aadu, 0x283dac3bu, 0x2d9cc348u,
                                0x2ce144adu, 0x21dd2e11u, 0x22eb35d7u, 0x2e9ed90bu, 0x29a692c4u, 0x27a8e5fdu, 0x00068042u};
    constexpr uint32_t P8[N] = {0x2ffd5558u, 0x2f7ffffdu, 0x2ffffdcdu, 0x2fff58a7u, 0x2120f55du, 0x207b5879u, 0x2b398693u,
                                0x29c2895du, 0x23ba5c25u, 0x25d66bb0u, 0x2d3db218u, 0x234d258bu, 0x2f51cbfdu, 0x000d0086u};
    constexpr uint32_t P16[N] = {0x2ffaaab0u, 0x2efffffdu, 0x2ffffb9du, 0x2ffeb151u, 0x2241eabdu, 0x20f6b0f4u, 0x26730d28u,
                                 0x238512bdu, 0x2774b84du, 0x2bacd762u, 0x2a7b6432u, 0x269a4b19u, 0x2ea397fcu, 0x001a010fu};
    constexpr uint32_t P32[N] = {0x2ff55560u, 0x2dfffffdu, 0x2ffff73du, 0x2ffd62a5u, 0x2483d57du, 0x21ed61eau, 0x2ce61a52u,
                                 0x270a257cu, 0x2ee9709cu, 0x2759aec6u, 0x24f6c867u, 0x2d349635u, 0x2d472ffau, 0x00340221u};
    return K == 2 ? P2[i] : K == 4 ? P4[i] : K == 8 ? P8[i] : K == 16 ? P16[i] : P32[i];
  }

  HD static Fp28 zero() {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = 0;
    return r;
  }
  HD static Fp28 one() {   // R' mod p
    constexpr uint32_t V[N] = {0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu,
                               0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au};
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = V[i];
    return r;
  }
  HD bool is_zero_limbs() const {   // exact zero representation (identity marker)
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc |= l[i];
    return acc == 0;
  }

  HD void normalize() {   // carry propagation only; value unchanged
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      l[i + 1] += l[i] >> B;
      l[i] &= MASK;
    }
  }

  // Montgomery product a*b/R' ; values bounded by a_k*p, b_k*p with a_k*b_k < 2528 ; result
  // normalised and < 2p.  Operands may be LAZY (limbs < 2^30, i.e. the un-normalised output of
  // sub_lazy / add_lazy): a column accumulator receives at most 14 products a_j*b_i < 2^60 and
  // 14 products m*p_j < 2^56 during its life, 14 * (2^60 + 2^56) < 2^63.9.
  HD static Fp28 mul(const Fp28& a, const Fp28& b) {
    uint64_t acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t bi = b.l[i];
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)a.l[j] * bi;
      const uint32_t m = ((uint32_t)acc[0] * INV) & MASK;
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)m * mod(j);
      const uint64_t carry = acc[0] >> B;   // low 28 bits are zero now
#pragma unroll
      for (int j = 0; j < N - 1; ++j) acc[j] = acc[j + 1];
      acc[N - 1] = 0;
      acc[0] += carry;
    }
    Fp28 r;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      acc[j + 1] += acc[j] >> B;
      r.l[j] = (uint32_t)acc[j] & MASK;
    }
    r.l[N - 1] = (uint32_t)acc[N - 1];
    return r;
  }
  // a*a/R' with the symmetric half of the partial products (105 instead of 196 mads): row i
  // adds a_i^2 and the doubled products 2 a_i a_j (j > i) into the columns i + j; column i is
  // complete when row i reduces it because every pair (k, i - k) was added by row min(k, i - k).
  // Same column totals as mul(a, a), hence the same bounds (operand may be lazy).
  HD Fp28 sqr() const {
    uint64_t acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t ai = l[i], ai2 = l[i] << 1;
      acc[i] += (uint64_t)ai * ai;
#pragma unroll
      for (int j = i + 1; j < N; ++j) acc[j] += (uint64_t)l[j] * ai2;
      // rows shift the window down by one column per iteration: position j holds column i + j,
      // so the products above were placed at positions (i + j) - i = j  [a_i^2 -> 2i - i = i]
      const uint32_t m = ((uint32_t)acc[0] * INV) & MASK;
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)m * mod(j);
      const uint64_t carry = acc[0] >> B;
#pragma unroll
      for (int j = 0; j < N - 1; ++j) acc[j] = acc[j + 1];
      acc[N - 1] = 0;
      acc[0] += carry;
    }
    Fp28 r;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      acc[j + 1] += acc[j] >> B;
      r.l[j] = (uint32_t)acc[j] & MASK;
    }
    r.l[N - 1] = (uint32_t)acc[N - 1];
    return r;
  }
  // (a*b + c*d)/R' with ONE Montgomery reduction.  Limb bounds: at most one operand of each
  // product lazy (limbs < 2^30), the other normalised: 14 * (2^58 + 2^58 + 2^56) < 2^63.
  // Value bounds: a_k*b_k + c_k*d_k < 2528 ; result normalised and < 2p.
  HD static Fp28 mul2(const Fp28& a, const Fp28& b, const Fp28& c, const Fp28& d) {
    uint64_t acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t bi = b.l[i], di = d.l[i];
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)a.l[j] * bi;
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)c.l[j] * di;
      const uint32_t m = ((uint32_t)acc[0] * INV) & MASK;
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)m * mod(j);
      const uint64_t carry = acc[0] >> B;
#pragma unroll
      for (int j = 0; j < N - 1; ++j) acc[j] = acc[j + 1];
      acc[N - 1] = 0;
      acc[0] += carry;
    }
    Fp28 r;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      acc[j + 1] += acc[j] >> B;
      r.l[j] = (uint32_t)acc[j] & MASK;
    }
    r.l[N - 1] = (uint32_t)acc[N - 1];
    return r;
  }

  // value(a) + value(b), normalised
  HD static Fp28 add(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + b.l[i];
    r.normalize();
    return r;
  }
  HD Fp28 dbl() const { return add(*this, *this); }
  // value(a) - value(b) + K p, normalised; requires b normalised and value(b) < (K/2) p
  template <int K>
  HD static Fp28 sub(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + (pad<K>(i) - b.l[i]);
    r.normalize();
    return r;
  }

  // ---- lazy forms: no carry propagation; limbs < 2^30 as long as the inputs are normalised.
  // Valid as operands of mul / sqr / mul2 (see their limb bounds), as the minuend of sub<K>,
  // and (add_lazy of two normalised values, limbs <= 2^29 - 2) as its subtrahend.
  HD static Fp28 add_lazy(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
  }
  template <int K>
  HD static Fp28 sub_lazy(const Fp28& a, const Fp28& b) {   // a, b normalised, value(b) < (K/2) p
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + (pad<K>(i) - b.l[i]);
    return r;
  }
  template <int K>
  HD static Fp28 neg_lazy(const Fp28& b) {                  // K p - value(b)
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = pad<K>(i) - b.l[i];
    return r;
  }
  HD Fp28 normalized() const {
    Fp28 r = *this;
    r.normalize();
    return r;
  }

  // canonical representative in [0, p); requires value < 64 p (conditional subtraction of 32p .. p)
  HD Fp28 canon() const {
    Fp28 r = *this;
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      uint32_t pl[N];   // limbs of 2^k p (top limb keeps the overflow bits)
      uint64_t c = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const uint64_t v = ((uint64_t)mod(i) << k) + c;
        pl[i] = (i == N - 1) ? (uint32_t)v : ((uint32_t)v & MASK);
        c = v >> B;
      }
      Fp28 t;
      int64_t borrow = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int64_t d = (int64_t)r.l[i] - (int64_t)pl[i] + borrow;
        if (i < N - 1) {
          borrow = d >> B;   // arithmetic shift: -1 when d < 0
          t.l[i] = (uint32_t)d & MASK;
        } else {
          borrow = d < 0 ? -1 : 0;
          t.l[i] = (uint32_t)d;
        }
      }
      if (borrow == 0) r = t;
    }
    return r;
  }
  HD bool is_zero_mod() const { return canon().is_zero_limbs(); }
  HD bool eq_mod(const Fp28& o) const { return sub<32>(*this, o.canon()).is_zero_mod(); }

  // ---- interop with the 12 x 32-bit, R = 2^384 form --------------------------------------
  HD static Fp28 reslice_from32(const uint32_t* w /*12*/) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int bit = B * i;
      const int wi = bit >> 5, sh = bit & 31;
      uint64_t v = (wi < 12 ? (uint64_t)w[wi] : 0) | ((wi + 1 < 12 ? (uint64_t)w[wi + 1] : 0) << 32);
      r.l[i] = (uint32_t)(v >> sh) & MASK;
    }
    return r;
  }
  HD void reslice_to32(uint32_t* w /*12*/) const {   // requires canonical (< 2^384)
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int bit = B * i;
      const int wi = bit >> 5, sh = bit & 31;
      const uint64_t v = (uint64_t)l[i] << sh;
      if (wi < 12) w[wi] |= (uint32_t)v;
      if (wi + 1 < 12) w[wi + 1] |= (uint32_t)(v >> 32);
    }
  }
  // x*2^384 (Fp)  ->  x*R' (Fp28): one product with 2^400 mod p
  HD static Fp28 from_fp(const Fp& x) {
    constexpr uint32_t C[N] = {0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u,
                               0xa4e6fe9u, 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau};
    Fp28 c;
#pragma unroll
    for (int i = 0; i < N; ++i) c.l[i] = C[i];
    return mul(reslice_from32(x.l), c);
  }
  // x*R' -> x*2^384, canonical: one product with 2^384 mod p
  HD Fp to_fp() const {
    constexpr uint32_t C[N] = {0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u,
                               0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u};
    Fp28 c;
#pragma unroll
    for (int i = 0; i < N; ++i) c.l[i] = C[i];
    const Fp28 t = mul(*this, c).canon();
    Fp r;
    t.reslice_to32(r.l);
    return r;
  }
};

}  // namespace plonk
