// Scalar -> table entries: the two digit recodings of the MSM (msm_sort.hip), host- and device-callable so that
// tests/test_field_host.py checks them on the CPU against the big-int model (tests/msm_wide_model.py).
//
// Both produce at most MSM_W = 16 non-zero digits per scalar s < 2^255, s = sum_j d_j * 2^(row_j), and call
//     f(slot j, row_j, bucket, sign)
// for each of them in order of increasing row; a table entry T[row][i] = 2^row * P_i (bit-position tables) or
// 2^(16 row) * P_i (window tables) then contributes sign * weight(bucket) * T[row][i].
//
//   * window tables (16 rows): signed 16-bit windows, d in [-2^15, 2^15], bucket = |d| - 1, weight = bucket + 1.
//     16 digits per scalar (a digit is zero with probability 2^-16).
//   * bit-position tables (256 rows, round 3): width-17 non-adjacent form.  A digit may start at ANY bit, so it is taken
//     where the remaining value is odd: d odd, |d| < 2^16, bucket = |d| >> 1 — the SAME 2^15 buckets hold 17-bit
//     digits, weight = 2 * bucket + 1, and after a digit the next 16 bits are zero: consecutive digits are >= 17 bits
//     apart and the expected distance is 18 (the run of equal bits after a digit has mean length 1): 254.9 / 18 + 1/2 =
//     14.7 additions per scalar instead of 16 (measured over random scalars: 14.67).  The price is a table row per bit
//     position — 256 x 128 B per point, 32 GiB at 2^20 points — which is what 288 GB of HBM is for (profiles/r03a/gather_tlb.txt: the random 128-B
//     gathers of msm_accumulate run at the same rate over 2, 34 or 137 GiB of tables).
#pragma once
#include "field.cuh"

namespace plonk {

static constexpr int MSM_DIGITS = 16;          // most non-zero digits of a scalar under either recoding (= MSM_W)
static constexpr uint32_t MSM_ROWS_WINDOW = 16, MSM_ROWS_BITPOS = 256, MSM_ROWS_HALFPOS = 128;
static constexpr uint32_t MSM_NAF_W = 17;      // digit width of the bit-position recoding over 2^15 buckets: odd |d| < 2^16

// Signed 16-bit windows, least significant first (carry into the next window when the value exceeds 2^15).
template <class S, class F>
HD void for_each_digit_window(const S& s, F&& f) {
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < MSM_DIGITS; ++w) {
    const uint32_t raw = (s.l[w >> 1] >> ((w & 1) * 16)) & 0xffffu;
    const uint32_t v = raw + carry;
    carry = 0;
    if (v > 32768u) {            // negative digit d = v - 65536
      carry = 1;
      const uint32_t mag = 65536u - v;
      if (mag) f(w, (uint32_t)w, mag - 1, 1u);
    } else if (v) {
      f(w, (uint32_t)w, v - 1, 0u);
    }
  }
}

// limb k of an 8 x 32-bit integer with a run-time k; k >= 8 -> 0.  Registers cannot be indexed at run time: a scalar
// held in registers goes through a select chain (8 compares + 8 selects per limb, ~70 instructions per digit — fine for
// the host and for tests); the kernels park the scalar in LDS instead (StridedLimbs: limb k of this lane's scalar at
// base[k * stride], one ds_read per limb, ~20 instructions per digit; r03: the select chains cost ~1 ms per proof).
template <class S>
HD uint32_t limb_select(const S& s, uint32_t k) {
  uint32_t r = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) r = (k == j) ? s.l[j] : r;
  return r;
}
struct StridedLimbs {     // NINE limbs are parked: limb 8 = 0, so that the 64-bit window at bit p < 256 needs no bounds test
  const uint32_t* base;   // limb 0 of the scalar
  uint32_t stride;        // words between consecutive limbs
};
HD uint32_t limb_select(const StridedLimbs& s, uint32_t k) { return s.base[k * s.stride]; }   // k <= 8
// bits [p, p + 32) of the 256-bit integer (zero beyond bit 255)
template <class S>
HD uint32_t bits32_at(const S& s, uint32_t p) {
  const uint32_t k = p >> 5, sh = p & 31;
  const uint64_t v = ((uint64_t)limb_select(s, k + 1) << 32) | limb_select(s, k);
  return (uint32_t)(v >> sh);
}

// Width-17 NAF of a canonical scalar (s < 2^255), least significant digit first.  `carry` = 1 after a negative
// digit: the value that remains above the digit is (s >> p) + carry, and it is odd exactly where bit p differs from the
// carry.  The top digit starts at bit 255 at the latest (s < 2^255 stops every carry there), so rows 0..255 suffice.
// W = digit width: odd digits |d| < 2^(W-1), bucket = |d| >> 1 in [0, 2^(W-2)); W = 17 for 2^15 buckets, 21 for 2^19.
template <uint32_t W, class S, class F>
HD void for_each_digit_naf(const S& s, F&& f) {
  uint32_t p = 0, carry = 0;
#pragma unroll
  for (int j = 0; j < MSM_DIGITS; ++j) {
    const uint32_t flip = 0u - carry;                  // all ones after a negative digit
    while (p < 256) {                                  // next bit that differs from the carry
      const uint32_t x = bits32_at(s, p) ^ flip;
      if (x) { p += (uint32_t)__builtin_ctz(x); break; }
      p += 32;
    }
    if (p >= 256) break;
    // Width of this digit.  Left alone, the LAST digit of a scalar is whatever remains above the last full one: 0 .. W - 1
    // bits, i.e. small with probability ~ 1 / value — the lowest ~100 buckets of every MSM would hold 5 % of the entries
    // (bucket 0 alone 10^5 of 1.2 * 10^7 at 2^20 terms).  So when at most two digits are left (256 - p <= 2 W bits remain,
    // s < 2^255) they share the remaining bits evenly: two digits of 11 .. 21 bits instead of 21 + (0 .. 20).
    const uint32_t rem = 256u - p;                                      // bits from p up, counting one bit of headroom above bit 254:
                                                                        // the last digit then never comes out negative (no +1 at row 255)
    const uint32_t w = (rem > W && rem <= 2 * W) ? (rem + 1) / 2 : W;
    const uint32_t v = (bits32_at(s, p) & ((1u << w) - 1u)) + carry;   // odd, < 2^w
    const uint32_t neg = v >> (w - 1);                                 // v > 2^(w-1): the digit is v - 2^w
    const uint32_t mag = neg ? (1u << w) - v : v;                      // odd, < 2^(w-1)
    f(j, p, mag >> 1, neg);
    carry = neg;
    p += w;
  }
}
template <class S, class F>
HD void for_each_digit_bitpos(const S& s, F&& f) { for_each_digit_naf<MSM_NAF_W>(s, f); }

// Half-density tables (round 4): a row for every EVEN bit position, T[r][i] = 2^(2 r) * P_i, r < 128 — half the memory of
// the bit-position tables (16 KiB per point), for keys whose 256 rows do not fit beside everything else (the Lagrange-basis
// key of a 2^22-gate circuit next to the commit key's 137 GB).  A digit may start at even positions only: it is taken where
// the remaining value (s >> p) + carry is NOT a multiple of 4, W bits wide (W even), signed: d in [-2^(W-1), 2^(W-1)],
// d != 0 mod 4, so digits are odd or 2 * odd and the bucket is |d| - 1 with weight bucket + 1 — the WINDOW convention (no
// "2 W - S").  After a digit the next W bits are consumed and the run of 2-bit groups equal to the carry that follows has
// mean length 1/3: 254.9 / (W + 2/3) + 1/2 digits per scalar — 12.8 for W = 20 over 2^19 buckets (a row per bit: 12.1;
// window rows: 16), 15.8 for W = 16 over 2^15.  The remaining value is a multiple of 4 exactly where the 2-bit group at p
// equals (carry, carry).  As in the NAF above the last two digits share the remaining bits (even widths), and no digit
// starts above bit 254: a digit at p >= 256 - W has no bit above 254 to read (s < 2^255), is positive and leaves no carry.
template <uint32_t W, class S, class F>
HD void for_each_digit_even(const S& s, F&& f) {
  static_assert(W % 2 == 0 && W >= 4 && W <= 22, "even digit width");
  uint32_t p = 0, carry = 0;
#pragma unroll
  for (int j = 0; j < MSM_DIGITS; ++j) {
    const uint32_t flip = 0u - carry;
    while (p < 256) {                                  // next 2-bit group that differs from (carry, carry)
      const uint32_t x = bits32_at(s, p) ^ flip;
      if (x) { p += (uint32_t)__builtin_ctz(x) & ~1u; break; }
      p += 32;
    }
    if (p >= 256) break;
    const uint32_t rem = 256u - p;
    const uint32_t w = (rem > W && rem <= 2 * W) ? ((rem / 2 + 1) & ~1u) : W;     // even, >= rem / 2
    const uint32_t v = (bits32_at(s, p) & ((1u << w) - 1u)) + carry;   // in [1, 2^w), not a multiple of 4
    const uint32_t neg = v > (1u << (w - 1)) ? 1u : 0u;                // the digit is v - 2^w
    const uint32_t mag = neg ? (1u << w) - v : v;                      // in [1, 2^(w-1)]
    f(j, p >> 1, mag - 1u, neg);
    carry = neg;
    p += w;
  }
}

template <class S, class F>
HD void for_each_digit(const S& s, uint32_t rows, F&& f) {
  if (rows == MSM_ROWS_BITPOS) for_each_digit_bitpos(s, f);
  else if (rows == MSM_ROWS_HALFPOS) for_each_digit_even<16>(s, f);
  else for_each_digit_window(s, f);
}

}  // namespace plonk
