// G1Affine::from_bytes for the 48-byte compressed encoding (the zcash / dusk-bls12_381 format the reference writes with
// G1Affine::to_bytes: big-endian x, flag bits 0x80 compressed | 0x40 infinity | 0x20 y is the lexicographically larger
// root) -> raw affine coordinates in Montgomery form, i.e. the first 96 bytes of G1Affine::to_raw_bytes.
//
// Used by the compressed commit-key loader (CommitKey::from_slice, reference src/commitment_scheme/kzg10/key.rs:319-326:
// one G1Affine::from_slice per 48-byte chunk).  One lane per point on the device; the same function compiles for the
// host (tests/csrc/host_arith.cpp checks it against the oracle).  The curve membership falls out of the square root;
// the prime-order-subgroup half of from_bytes is srs_validate_kernel's job (msm.hip) on the decoded points.
#pragma once
#include "curve.cuh"

namespace plonk {

enum : int { G1DEC_OK = 0, G1DEC_INVALID = 1, G1DEC_IDENTITY = 2 };

HD int g1_decompress48(const uint8_t* in, G1Affine* out) {
  const uint8_t flags = in[0];
  if (!(flags & 0x80)) return G1DEC_INVALID;                     // uncompressed encodings are a different (96-byte) format
  Fp x;                                                          // the integer x, little-endian words
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pos = 47 - 4 * k - b;
      const uint32_t v = pos == 0 ? (uint32_t)(in[0] & 0x1f) : (uint32_t)in[pos];
      w |= v << (8 * b);
    }
    x.l[k] = w;
  }
  if (flags & 0x40) return (!(flags & 0x20) && x.is_zero()) ? G1DEC_IDENTITY : G1DEC_INVALID;
  bool lt = false;                                               // canonical: x < p
  for (int k = 11; k >= 0; --k)
    if (x.l[k] != FpP::MOD[k]) { lt = x.l[k] < FpP::MOD[k]; break; }
  if (!lt) return G1DEC_INVALID;
  const Fp xm = x.to_mont();
  const Fp y2 = xm.sqr() * xm + Fp::from_u64(4);
  uint32_t e[12];                                                // (p + 1) / 4   (p = 3 mod 4; p + 1 does not carry out of word 0)
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const uint32_t lo = FpP::MOD[k] + (k == 0 ? 1u : 0u);
    const uint32_t hi = k < 11 ? FpP::MOD[k + 1] : 0u;
    e[k] = (lo >> 2) | (hi << 30);
  }
  Fp y = y2.pow_words(e, 12);
  if (y.sqr() != y2) return G1DEC_INVALID;                       // x^3 + 4 is not a square: no such point
  const Fp yi = y.from_mont();                                   // the integer y: larger root  <=>  y > (p - 1) / 2
  bool larger = false;
  for (int k = 11; k >= 0; --k) {
    const uint32_t half = (FpP::MOD[k] >> 1) | (k < 11 ? FpP::MOD[k + 1] << 31 : 0u);   // (p - 1) / 2 = p >> 1 (p odd)
    if (yi.l[k] != half) { larger = yi.l[k] > half; break; }
  }
  if (larger != ((flags & 0x20) != 0)) y = y.neg();
  out->x = xm;
  out->y = y;
  return G1DEC_OK;
}

}  // namespace plonk
