// A 16-lane COOPERATIVE Montgomery product over BLS12-381 Fp (fp28.cuh's form: 14 limbs of 28 bits, R' = 2^392) — the
// prototype DESIGN.md section 7 item 2 sizes: one DPP row of 16 lanes computes ONE product, so that a lone wave in the
// latency-bound tail of an MSM (row / column sums, bit sums, fold) waits ~1/4 as long for it as for the 490 dependent
// instructions of Fp28::mul on a single lane.  Measurement prototype only (tools/ubench/coop_mul.hip); no product code uses it.
//
// Data layout.  `a`: UNIFORM over the row (14 limbs known to every lane — kernel arguments / v_readlane results live in
// SGPRs, which v_mad_u64_u32 takes as a source).  `b` and the result: DISTRIBUTED — lane j of the row holds limb j
// (j < 14), lanes 14 and 15 hold 0.  All limbs may be "lazy" (< 2^30): 14 * (2^30)^2 < 2^64.
//
// Columns.  lane k accumulates   lo_k = sum_i a_i * b_(k-i)      = column k        (k = 0 .. 15)
//                                hi_k = sum_i a_i * b_(k+16-i)   = column k + 16   (k = 0 .. 10)
// with b_(k-i) = row_shr:i of the distributed b and b_(k+16-i) = row_shl:(16-i) — the zero fill of a DPP shift IS the
// range mask, so a product is 27 v_mov_dpp + 28 v_mad_u64_u32 and no select.
//
// Reduction (Montgomery, R' = 2^392 = columns 0 .. 13) as two more products by CONSTANTS, with carries handled lazily:
//   1. near-normalise columns 0 .. 13: a column < 2^64 is cut into 28 + 28 + 8 bits, the upper pieces go one / two lanes
//      up (2 DPP shifts): C'_k < 2^30, VALUE unchanged; what leaves column 13 lands in lanes 14 / 15 (columns 14 / 15).
//   2. m = C' * (-p^-1) mod 2^392: columns 0 .. 13 of that product (14 + 14), near-normalised the same way, pieces beyond
//      column 13 dropped (that is "mod 2^392").  m' is REDUNDANT (limbs < 2^30, value < 4 * 2^392): c + m' p is still a
//      multiple of 2^392, the result just grows by at most 4p — no exact limbs, no carry ripple anywhere.
//   3. t = m' * p (27 + 28), u = c + t column by column.
//   4. the low columns of u sum to e * 2^392 EXACTLY with e what has to be carried into column 14: two shift-and-add passes
//      bring every low column under 2^28 + 2^8 while moving the bulk of e into lane 14; what is left in lanes 0 .. 13 is
//      then either 0 or exactly 2^392, and it is 2^392 iff lane 13 is non-zero — one flag, one DPP shift.
//   5. result limb j = column 14 + j: lanes 14, 15 (lo) -> lanes 0, 1 (row_shl:14), hi_k -> lane k + 2 (row_shr:2), then
//      one near-normalisation: limbs < 2^30, value < a b / 2^392 + 4p.
//
// The algorithm is written ONCE over a backend B (B::u32 / B::u64 = a value per lane): tools/ubench/coop_mul.hip
// instantiates it with DPP intrinsics, tests/csrc/host_arith.cpp with 16-element arrays (the CPU test checks the carry
// logic against Fp28::mul bit for bit modulo p).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define COOP_HD __host__ __device__ __forceinline__
#else
#define COOP_HD inline
#endif

namespace coop {

static constexpr int N = 14;
static constexpr uint32_t MASK = (1u << 28) - 1;
// the uniform operand of a product: something with at(i), i a compile-time constant after unrolling
struct PLimbs {    // p, 28-bit limbs
  COOP_HD static constexpr uint32_t at(int i) {
    constexpr uint32_t M[N] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                               0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
    return M[i];
  }
};
struct NPLimbs {   // -p^-1 mod 2^392
  COOP_HD static constexpr uint32_t at(int i) {
    constexpr uint32_t M[N] = {0xffcfffdu, 0xf3fffcfu, 0x113e889u, 0xdb92d9du, 0xb48286au, 0xf0c8e30u, 0xc16ef2eu,
                               0x8eb2db4u, 0x9ecca0eu, 0x68cf581u, 0x316fee2u, 0xfc9468bu, 0x106feaau, 0xa0ceb06u};
    return M[i];
  }
};
struct Uniform {   // 14 limbs every lane knows (kernel arguments: SGPRs)
  uint32_t l[N];
  COOP_HD uint32_t at(int i) const { return l[i]; }
};

template <class B>
struct Mul {
  using u32 = typename B::u32;
  using u64 = typename B::u64;

  // columns of (uniform a) x (distributed b); HI = false: only lo (columns 0 .. 15)
  template <bool HI, int I, class A>
  static COOP_HD void rows(const A& a, const u32& b, u64& lo, u64& hi) {
    if constexpr (I < N) {
      lo = B::mad(lo, a.at(I), B::template shr32<I>(b));
      if constexpr (HI && I >= 1) hi = B::mad(hi, a.at(I), B::template shl32<16 - I>(b));
      rows<HI, I + 1>(a, b, lo, hi);
    }
  }
  // a 64-bit column cut into 28 + 28 + 8 bits, the upper pieces moved one / two lanes up.  Lanes >= 14 are NOT cut (they hold
  // whole columns 14 / 15 and collect what leaves column 13) when KEEP_TOP, else they are cleared (a 14-limb value).
  template <bool KEEP_TOP>
  static COOP_HD u64 near_normalise(const u64& c) {
    const u32 low = B::lane_lt(14);
    const u64 cut = B::select64(low, c, B::zero64());
    const u32 p0 = B::and32(B::lo32(cut), MASK);
    const u32 p1 = B::and32(B::lo32(B::shr64_bits(cut, 28)), MASK);
    const u32 p2 = B::lo32(B::shr64_bits(cut, 56));
    u64 r = B::add64(B::widen(p0), B::widen(B::add32(B::template shr32<1>(p1), B::template shr32<2>(p2))));
    if constexpr (KEEP_TOP) r = B::add64(r, B::select64(low, B::zero64(), c));
    else r = B::select64(low, r, B::zero64());
    return r;
  }

  // a: 14 uniform limbs (< 2^30), b: distributed (< 2^30, lanes 14 / 15 zero) -> a b / 2^392 mod p, distributed, limbs < 2^30
  static COOP_HD u32 mul(const Uniform& a, const u32& b) {
    u64 lo = B::zero64(), hi = B::zero64();
    rows<true, 0>(a, b, lo, hi);
    // 1. C': columns 0 .. 13 as limbs < 2^30, columns 14 / 15 whole (+ what left column 13)
    const u64 c1 = near_normalise<true>(lo);
    const u32 cd = B::lo32(c1);                       // lanes 0 .. 13: C'_k (lanes 14 / 15 never reach a column <= 13 below)
    // 2. m' = C' * (-p^-1) mod 2^392
    u64 mlo = B::zero64(), unused = B::zero64();
    rows<false, 0>(NPLimbs{}, cd, mlo, unused);
    const u32 md = B::lo32(near_normalise<false>(mlo));
    // 3. u = c + m' p
    u64 tlo = B::zero64(), thi = B::zero64();
    rows<true, 0>(PLimbs{}, md, tlo, thi);
    const u64 ulo = B::add64(c1, tlo), uhi = B::add64(hi, thi);
    // 4. carry out of the low columns: two shift-and-add passes, then the flag of lane 13
    const u32 low = B::lane_lt(14);
    u64 v = ulo;
    for (int pass = 0; pass < 2; ++pass) {
      const u64 cut = B::select64(low, v, B::zero64());
      const u64 h = B::shr64_bits(cut, 28);
      const u64 keep = B::select64(low, B::widen(B::and32(B::lo32(cut), MASK)), v);
      v = B::add64(keep, B::template shr64<1>(h));
    }
    const u32 flag = B::and32(B::lane_eq(13), B::nonzero32(B::lo32(v)));    // all-ones mask in lane 13 iff its limb is non-zero
    v = B::add64(v, B::widen(B::and32(B::template shr32<1>(flag), 1u)));    // lane 14: + 1
    // 5. result limb j = column 14 + j
    const u64 top = B::select64(low, B::zero64(), v);                       // lanes 14, 15
    const u64 r = B::add64(B::template shl64<14>(top), B::template shr64<2>(uhi));
    return B::lo32(near_normalise<false>(r));
  }
};

}  // namespace coop
