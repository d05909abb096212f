// BLS12-381 Fr in reduced radix for the NTT butterflies: 9 limbs of 29 bits, Montgomery
// reduction with R'' = 2^261, Harvey-style lazy range [0, 2q + eps).
//
// Same rationale as fp28.cuh: on gfx950 v_mad_u64_u32 costs what an integer add costs, so the
// multiplier is built from 2*9*9 mads with 64-bit column accumulators (18 products of < 2^59
// fit) and no carry instructions.  Data in HBM keeps the reference's layout (8 x 32-bit limbs,
// Montgomery R = 2^256); only registers/LDS use this form.  Twiddles are stored as w * R''
// so that  mul(x*R, w*R'') = x*w*R  — the data never changes Montgomery domain.
//
// Ranges: loads are canonical (< q).  add_csub keeps sums < 2q + eps with a conditional
// subtraction of 2q decided on the top limbs only (eps <= 2^233 * 2^stages — irrelevant next to
// q ~ 2^255 within one pass of <= 9 stages); sub_lazy returns a - b + 4q (< 6q + eps) with
// limbs < 1.5 * 2^30, which mul() accepts as its first operand; mul() returns < 2q, normalised.
// R''/q = 70.6, so operand bounds (6 x 2) are far inside the Montgomery condition.
// to_fr() is exact (canonical), so every value written back to HBM is bit-identical to the
// reference's.
#pragma once
#include "field.cuh"

namespace plonk {

struct Fr29 {
  static constexpr int N = 9;
  static constexpr int B = 29;
  static constexpr uint32_t MASK = (1u << B) - 1;
  uint32_t l[N];

  HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t M[N] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu,
                               0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
    return M[i];
  }
  HD static constexpr uint32_t neg2q(int i) {   // 2^261 - 2q
    constexpr uint32_t V[N] = {0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u,
                               0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x1f1824b1u};
    return V[i];
  }
  HD static constexpr uint32_t pad4(int i) {    // 4q with limbs in [2^29, 2^30)
    constexpr uint32_t V[N] = {0x20000004u, 0x3fffffdfu, 0x3e5bfefeu, 0x2d2017feu, 0x360154eeu,
                               0x30101342u, 0x3483339cu, 0x2994cebdu, 0x01cfb69cu};
    return V[i];
  }
  HD static constexpr uint32_t neg4q(int i) {   // 2^261 - 4q
    constexpr uint32_t V[N] = {0x1ffffffcu, 0x0000001fu, 0x01a40100u, 0x12dfe800u, 0x09feab10u,
                               0x0fefecbcu, 0x0b7ccc62u, 0x166b3141u, 0x1e304962u};
    return V[i];
  }
  static constexpr uint32_t TOP_2Q = 0xe7db4eu;   // (2q) >> 232
  static constexpr uint32_t TOP_4Q = 0x1cfb69du;  // (4q) >> 232

  HD static Fr29 zero() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = 0;
    return r;
  }
  HD void normalize() {
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      l[i + 1] += l[i] >> B;
      l[i] &= MASK;
    }
  }

  // Montgomery product a*w/R''.  a: limbs < 1.5*2^30, value < ~8q.  w: normalised, value < 2q.
  // Result normalised, < 2q.
  HD static Fr29 mul(const Fr29& a, const Fr29& w) {
    uint64_t acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t wi = w.l[i];
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)a.l[j] * wi;
      const uint32_t m = (0u - (uint32_t)acc[0]) & MASK;   // -q^-1 = 2^29 - 1 (q = 1 mod 2^32)
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += (uint64_t)m * mod(j);
      const uint64_t carry = acc[0] >> B;
#pragma unroll
      for (int j = 0; j < N - 1; ++j) acc[j] = acc[j + 1];
      acc[N - 1] = 0;
      acc[0] += carry;
    }
    Fr29 r;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      acc[j + 1] += acc[j] >> B;
      r.l[j] = (uint32_t)acc[j] & MASK;
    }
    r.l[N - 1] = (uint32_t)acc[N - 1];
    return r;
  }

  // a + b, minus 2q when the top limbs say the sum exceeds 2q; normalised.
  HD static Fr29 add_csub(const Fr29& a, const Fr29& b) {
    const uint32_t sel = (a.l[N - 1] + b.l[N - 1] > TOP_2Q) ? 0xffffffffu : 0u;
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + b.l[i] + (neg2q(i) & sel);
    r.normalize();
    r.l[N - 1] &= MASK;   // drops the 2^261 that came with neg2q
    return r;
  }
  // a - b + 4q, NOT normalised (limbs < 1.5 * 2^30): feed to mul() as first operand.
  HD static Fr29 sub_lazy(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + (pad4(i) - b.l[i]);
    return r;
  }

  // a - b reduced to < 2q + eps without a multiplication (the twiddle-free last DIF stage):
  // a - b + 4q < 6q, then top-limb-guided subtractions of 4q and 2q.  Normalised.
  HD static Fr29 sub_reduce(const Fr29& a, const Fr29& b) {
    Fr29 r = sub_lazy(a, b);
    r.normalize();
    const uint32_t s4 = (r.l[N - 1] > TOP_4Q) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] += neg4q(i) & s4;
    r.normalize();
    r.l[N - 1] &= MASK;
    const uint32_t s2 = (r.l[N - 1] > TOP_2Q) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] += neg2q(i) & s2;
    r.normalize();
    r.l[N - 1] &= MASK;
    return r;
  }

  // exact conditional subtraction of q
  HD Fr29 csub_q() const {
    Fr29 t;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int32_t d = (int32_t)l[i] - (int32_t)mod(i) + borrow;
      borrow = d >> 31;   // -1 when negative (|d| < 2^30)
      t.l[i] = (i < N - 1) ? ((uint32_t)d & MASK) : (uint32_t)d;
    }
    return borrow ? *this : t;
  }
  // ---- HBM format: 8 x 32-bit limbs, canonical, Montgomery R = 2^256 ------------------
  HD static Fr29 from_fr(const Fr& x) {   // pure re-slicing
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int bit = B * i;
      const int wi = bit >> 5, sh = bit & 31;
      const uint64_t v = (uint64_t)x.l[wi] | ((wi + 1 < 8 ? (uint64_t)x.l[wi + 1] : 0) << 32);
      r.l[i] = (uint32_t)(v >> sh) & MASK;
    }
    return r;
  }
  // normalised value < 4q -> canonical Fr
  HD Fr to_fr() const {
    // < 4q: subtract 2q if possible, then q
    Fr29 t = *this;
    {
      Fr29 u;
      int32_t borrow = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const uint32_t q2 = (i == 0) ? 2u : (((mod(i) << 1) | (mod(i - 1) >> (B - 1))) & ((i == N - 1) ? 0xffffffffu : MASK));
        const int32_t d = (int32_t)t.l[i] - (int32_t)q2 + borrow;
        borrow = d >> 31;
        u.l[i] = (i < N - 1) ? ((uint32_t)d & MASK) : (uint32_t)d;
      }
      if (!borrow) t = u;
    }
    t = t.csub_q();
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.l[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int bit = B * i;
      const int wi = bit >> 5, sh = bit & 31;
      const uint64_t v = (uint64_t)t.l[i] << sh;
      r.l[wi] |= (uint32_t)v;
      if (wi + 1 < 8) r.l[wi + 1] |= (uint32_t)(v >> 32);
    }
    return r;
  }
  // x*R (Fr) -> x*R'' : twiddle tables are kept in this form
  HD static Fr29 twiddle_from_fr(const Fr& x) {
    constexpr uint32_t C[N] = {0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u,
                               0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu};   // R''^2 / R mod q
    Fr29 c;
#pragma unroll
    for (int i = 0; i < N; ++i) c.l[i] = C[i];
    return mul(from_fr(x), c).csub_q();
  }
};

struct alignas(16) Fr29Slot {   // 9 limbs padded to 12 words (48 B) for aligned 16-byte loads
  uint32_t w[12];
};

}  // namespace plonk
