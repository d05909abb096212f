// Host-side Fiat-Shamir transcript: Merlin 3.0 (STROBE-128 over Keccak-f[1600]) and the
// reference's TranscriptProtocol wrapper (reference src/transcript.rs:90-145;
// VerifierKey::seed_transcript, src/proof_system/widget.rs:218-258).  Challenges decide
// the Proof bytes, so this is part of the product's host driver (written from the
// Merlin / STROBE specifications; the crates are external to the reference tree).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "field.cuh"

namespace plonk {

inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(uint8_t st[200]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
      0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
      0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
      0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
      0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61},
                                {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
  uint64_t a[5][5];
  for (int x = 0; x < 5; ++x)
    for (int y = 0; y < 5; ++y) memcpy(&a[x][y], st + 8 * (x + 5 * y), 8);
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint64_t c[5], d[5], b[5][5];
    for (int x = 0; x < 5; ++x) c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) a[x][y] ^= d[x];
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) b[y][(2 * x + 3 * y) % 5] = rotl64(a[x][y], ROT[x][y]);
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) a[x][y] = b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]);
    a[0][0] ^= RC[rnd];
  }
  for (int x = 0; x < 5; ++x)
    for (int y = 0; y < 5; ++y) memcpy(st + 8 * (x + 5 * y), &a[x][y], 8);
}

class Strobe128 {
 public:
  explicit Strobe128(const uint8_t* label, size_t len) {
    memset(st_, 0, sizeof st_);
    const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
    memcpy(st_, hdr, 6);
    memcpy(st_ + 6, "STROBEv1.0.2", 12);
    keccak_f1600(st_);
    meta_ad(label, len, false);
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
  void ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
  void prf(uint8_t* out, size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); squeeze(out, n); }

 private:
  static constexpr uint8_t R = 166;
  static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;
  uint8_t st_[200];
  uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
  void run_f() {
    st_[pos_] ^= pos_begin_;
    st_[pos_ + 1] ^= 0x04;
    st_[R + 1] ^= 0x80;
    keccak_f1600(st_);
    pos_ = 0;
    pos_begin_ = 0;
  }
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      st_[pos_] ^= d[i];
      if (++pos_ == R) run_f();
    }
  }
  void squeeze(uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      d[i] = st_[pos_];
      st_[pos_] = 0;
      if (++pos_ == R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;   // continuing the same operation
    const uint8_t old_begin = pos_begin_;
    pos_begin_ = pos_ + 1;
    cur_flags_ = flags;
    const uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
  }
};

// canonical 32-byte LE (BlsScalar::to_bytes) and wide reduction (from_bytes_wide)
inline void fr_to_bytes(const Fr& x, uint8_t out[32]) {
  const Fr c = x.from_mont();
  memcpy(out, c.l, 32);
}
inline Fr fr_from_bytes_wide(const uint8_t b[64]) {
  // value = lo + hi * 2^256 ; Montgomery(lo) = lo * R2 * R^-1 ; Montgomery(hi * 2^256) = hi * R2 (as Montgomery product with R2 twice)
  Fr lo, hi;
  memcpy(lo.l, b, 32);
  memcpy(hi.l, b + 32, 32);
  // lo, hi may exceed q (< 2^256): the Montgomery product tolerates inputs < 2^256 here
  // because q > 2^254: a*b*R^-1 with a < 2^256, b < q stays < 2q before the final subtraction.
  const Fr r2 = Fr::r2();
  Fr lo_m = lo * r2;            // lo * R
  Fr hi_m = (hi * r2) * r2;     // hi * R * R = (hi * 2^256) * R
  return lo_m + hi_m;
}

class Transcript {
 public:
  explicit Transcript(const uint8_t* label, size_t len) : s_((const uint8_t*)"Merlin v1.0", 11) {
    append_message("dom-sep", label, len);
  }
  void append_message(const char* label, const uint8_t* msg, size_t len) {
    const uint32_t l = (uint32_t)len;
    uint8_t le[4] = {(uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
    s_.meta_ad((const uint8_t*)label, strlen(label), false);
    s_.meta_ad(le, 4, true);
    s_.ad(msg, len, false);
  }
  void append_u64(const char* label, uint64_t x) {
    uint8_t le[8];
    for (int i = 0; i < 8; ++i) le[i] = (uint8_t)(x >> (8 * i));
    append_message(label, le, 8);
  }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) {
    const uint32_t l = (uint32_t)n;
    uint8_t le[4] = {(uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
    s_.meta_ad((const uint8_t*)label, strlen(label), false);
    s_.meta_ad(le, 4, true);
    s_.prf(out, n, false);
  }
  // TranscriptProtocol (transcript.rs:90-108)
  void append_commitment(const char* label, const uint8_t c48[48]) { append_message(label, c48, 48); }
  void append_scalar(const char* label, const Fr& s) {
    uint8_t b[32];
    fr_to_bytes(s, b);
    append_message(label, b, 32);
  }
  Fr challenge_scalar(const char* label) {
    uint8_t buf[64];
    challenge_bytes(label, buf, 64);
    return fr_from_bytes_wide(buf);
  }
  void circuit_domain_sep(uint64_t n) {
    append_message("dom-sep", (const uint8_t*)"circuit_size", 12);
    append_u64("n", n);
  }

 private:
  Strobe128 s_;
};

}  // namespace plonk
