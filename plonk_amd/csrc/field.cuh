// BLS12-381 Fr (8 x u32) and Fp (12 x u32) Montgomery arithmetic for gfx950.
//
// Storage is bit-identical to the reference types: BlsScalar.0 = 4 x u64 LE
// Montgomery limbs (R = 2^256; pinned by the MINUS_ONE literal at reference
// src/composer.rs:334-339) and Fp = 6 x u64 LE Montgomery limbs (R = 2^384;
// raw CommitKey form, reference src/commitment_scheme/kzg10/key.rs:215-229).
// A u64 LE limb is two u32 LE limbs, so no conversion is needed at the ABI.
//
// Everything is __host__ __device__ so the exact same code is unit-tested on
// the CPU (tests/test_field_host.py) against the big-int oracle.  This is
// 256/384-bit integer work: VALU v_mad_u64_u32 chains, no MFMA.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace plonk {

template <int N>
struct alignas(16) Big {
  uint32_t l[N];
};

// ---- field parameter packs -------------------------------------------------
struct FrP {
  static constexpr int N = 8;
  static constexpr uint32_t INV = 0xffffffffu;
  static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                      0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                      0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
  static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                     0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};
struct FpP {
  static constexpr int N = 12;
  static constexpr uint32_t INV = 0xfffcfffdu;
  static constexpr uint32_t MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                       0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                       0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                                       0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                                       0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
  static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                                      0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                                      0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};

// ---- generic Montgomery field ---------------------------------------------
template <class P>
struct Field {
  static constexpr int N = P::N;
  uint32_t l[N];

  HD static Field zero() {
    Field r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = 0;
    return r;
  }
  HD static Field one() {
    Field r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = P::ONE[i];
    return r;
  }
  HD static Field r2() {
    Field r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = P::R2[i];
    return r;
  }
  HD bool is_zero() const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc |= l[i];
    return acc == 0;
  }
  HD bool operator==(const Field& o) const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc |= l[i] ^ o.l[i];
    return acc == 0;
  }
  HD bool operator!=(const Field& o) const { return !(*this == o); }

  // r = a - MOD if a >= MOD (a < 2*MOD)
  HD static Field reduce_once(const Field& a, uint32_t top_carry = 0) {
    Field d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t t = (uint64_t)a.l[i] - P::MOD[i] - borrow;
      d.l[i] = (uint32_t)t;
      borrow = (t >> 63) & 1;
    }
    // keep the difference if no borrow, or if the sum overflowed 2^(32N)
    bool use = (borrow == 0) || (top_carry != 0);
    Field r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = use ? d.l[i] : a.l[i];
    return r;
  }

  HD friend Field operator+(const Field& a, const Field& b) {
    Field s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t t = (uint64_t)a.l[i] + b.l[i] + c;
      s.l[i] = (uint32_t)t;
      c = t >> 32;
    }
    return reduce_once(s, (uint32_t)c);
  }
  HD friend Field operator-(const Field& a, const Field& b) {
    Field d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
      d.l[i] = (uint32_t)t;
      borrow = (t >> 63) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;   // add MOD back if we borrowed
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t t = (uint64_t)d.l[i] + (P::MOD[i] & mask) + c;
      d.l[i] = (uint32_t)t;
      c = t >> 32;
    }
    return d;
  }
  HD Field neg() const { return zero() - *this; }
  HD Field dbl() const { return *this + *this; }

  // CIOS Montgomery product: a*b*R^-1 mod MOD, fully reduced.
  HD friend Field operator*(const Field& a, const Field& b) {
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t c = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        uint64_t s = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
        t[j] = (uint32_t)s;
        c = s >> 32;
      }
      uint64_t s = (uint64_t)t[N] + c;
      t[N] = (uint32_t)s;
      t[N + 1] = (uint32_t)(s >> 32);
      uint32_t m = t[0] * P::INV;
      s = (uint64_t)m * P::MOD[0] + t[0];
      c = s >> 32;
#pragma unroll
      for (int j = 1; j < N; ++j) {
        s = (uint64_t)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (uint32_t)s;
        c = s >> 32;
      }
      s = (uint64_t)t[N] + c;
      t[N - 1] = (uint32_t)s;
      t[N] = t[N + 1] + (uint32_t)(s >> 32);
    }
    Field r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = t[i];
    return reduce_once(r, t[N]);
  }
  HD Field sqr() const { return (*this) * (*this); }

  // canonical (non-Montgomery) limbs -> Montgomery and back
  HD Field to_mont() const { return (*this) * r2(); }
  HD Field from_mont() const {
    Field o = zero();
    o.l[0] = 1;
    return (*this) * o;
  }
  HD static Field from_u64(uint64_t v) {
    Field r = zero();
    r.l[0] = (uint32_t)v;
    r.l[1] = (uint32_t)(v >> 32);
    return r.to_mont();
  }

  // x^e, e given as little-endian u32 words (variable time; e is public)
  HD Field pow_words(const uint32_t* e, int nwords) const {
    Field acc = one();
    bool started = false;
    for (int w = nwords - 1; w >= 0; --w)
      for (int b = 31; b >= 0; --b) {
        if (started) acc = acc.sqr();
        if ((e[w] >> b) & 1) {
          acc = started ? acc * (*this) : *this;
          started = true;
        }
      }
    return acc;
  }
  HD Field pow_u64(uint64_t e) const {
    uint32_t w[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
    return pow_words(w, 2);
  }
  // Fermat inverse x^(MOD-2); 0 -> 0
  HD Field inv() const {
    uint32_t e[N];
    uint64_t borrow = 2;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t t = (uint64_t)P::MOD[i] - borrow;
      e[i] = (uint32_t)t;
      borrow = (t >> 63) & 1;
    }
    return pow_words(e, N);
  }
};

using Fr = Field<FrP>;
using Fp = Field<FpP>;

// constants as Montgomery Fr
HD Fr fr_generator() {   // GENERATOR = 7 (reference domain.rs:115)
  Fr r;
  const uint32_t v[8] = {0xfffffff1u, 0x0000000eu, 0x00189c0fu, 0x17e363d3u,
                         0x6f8457b0u, 0xff9c5787u, 0x8fc5a8c4u, 0x35133220u};
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = v[i];
  return r;
}
HD Fr fr_root_of_unity() {   // 7^((q-1)/2^32), order 2^32 (TWO_ADACITY = 32)
  Fr r;
  const uint32_t v[8] = {0x5f0e466au, 0xb9b58d8cu, 0x1819d7ecu, 0x5b1b4c80u,
                         0x52a31e64u, 0x0af53ae3u, 0x19e9b27bu, 0x5bf3addau};
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = v[i];
  return r;
}

}  // namespace plonk
