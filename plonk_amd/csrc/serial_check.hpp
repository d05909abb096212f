// The HIP-free half of serial.hip: the byte-level validation of a `Prover::to_bytes()` blob (blob_check) and of a
// PublicParameters file (public_parameters_check) — everything plonk_prover_blob_check / plonk_public_parameters_check do and
// the first step of plonk_prover_from_bytes / plonk_srs_load_public_parameters.  These functions read UNTRUSTED bytes, and row f4
// of SURVEY section 8 cannot be pinned to reference-produced files here (no cargo): they live in a header of their own so that
// tests/fuzz/fuzz_serial.cpp can compile exactly this code for the host under AddressSanitizer + UBSan and libFuzzer
// (round 6).  No HIP types, no allocation, no recursion; every offset is checked before it is used.
//
// Decoding mirrors, check for check:
//   Prover::try_from_bytes            src/compiler/prover.rs:266-345
//   ProverKey::from_slice             src/proof_system/widget.rs:449-625
//   Evaluations::from_slice           src/fft/evaluations.rs:64-90
//   EvaluationDomain::{from_bytes,new,matches_*}   src/fft/domain.rs:81-105,122-158,307-352
//   Polynomial::from_slice            src/fft/polynomial.rs:152-163
//   CommitKey::from_raw_var_bytes     src/commitment_scheme/kzg10/key.rs:263-300
//   VerifierKey (layout only)         src/proof_system/widget.rs:84-134
//   PublicParameters::{from_slice, from_slice_unchecked, trim}   src/commitment_scheme/kzg10/srs.rs:103-196
#pragma once
#include <cstdint>
#include <cstring>
#include <initializer_list>

#include "../../include/plonk_hip.h"
#include "hostg2.hpp"    // hostg1.hpp (g1_compressed_valid), curve28.cuh, field.cuh
#include "g1codec.cuh"

namespace plonk {
void set_last_error(const char* what, const char* detail, const char* file, int line);   // capi.hip (the fuzz harness brings its own)
namespace {

constexpr uint64_t SCALAR = 32, DOMAIN_BYTES = 8 + 4 + 5 * SCALAR, RAW_POINT = 97, VK_BYTES = 20 * 48 + 8;

inline uint64_t be64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 0; i < 8; ++i) v = (v << 8) | p[i];
  return v;
}
inline uint64_t le64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
  return v;
}
inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// BlsScalar::from_bytes: 32 bytes little-endian, must be < q
inline bool scalar_from_bytes(const uint8_t* b, Fr* out) {
  Fr raw;
  for (int i = 0; i < 8; ++i) raw.l[i] = le32(b + 4 * i);
  bool lt = false;
  for (int i = 7; i >= 0; --i) {
    if (raw.l[i] != FrP::MOD[i]) { lt = raw.l[i] < FrP::MOD[i]; break; }
  }
  if (!lt) return false;
  if (out) *out = raw.to_mont();
  return true;
}
inline bool scalars_canonical(const uint8_t* b, uint64_t count) {
  for (uint64_t k = 0; k < count; ++k)
    if (!scalar_from_bytes(b + SCALAR * k, nullptr)) return false;
  return true;
}

struct Domain {   // EvaluationDomain::new (domain.rs:122-158)
  uint64_t size = 0;
  uint32_t log = 0;
  Fr size_fe, size_inv, group_gen, group_gen_inv, generator_inv;
  bool init(uint64_t n) {
    if (n == 0 || (n & (n - 1))) return false;
    log = (uint32_t)__builtin_ctzll(n);
    if (log >= 32) return false;   // TWO_ADACITY
    size = n;
    group_gen = fr_root_of_unity();
    for (uint32_t i = log; i < 32; ++i) group_gen = group_gen.sqr();
    size_fe = Fr::from_u64(n);
    size_inv = size_fe.inv();
    group_gen_inv = group_gen.inv();
    generator_inv = fr_generator().inv();
    return true;
  }
  bool matches(const uint8_t* b) const {   // the 172 serialized bytes describe exactly this domain
    if (le64(b) != size || le32(b + 8) != log) return false;
    const Fr* want[5] = {&size_fe, &size_inv, &group_gen, &group_gen_inv, &generator_inv};
    for (int k = 0; k < 5; ++k) {
      Fr got;
      if (!scalar_from_bytes(b + 12 + SCALAR * k, &got) || got != *want[k]) return false;
    }
    return true;
  }
};

int fail(int code, const char* what, int line) {
  set_last_error(code == PLONK_ERR_BYTES ? "NotEnoughBytes" : code == PLONK_ERR_POINT ? "PointMalformed" : "InvalidData", what, __FILE__, line);
  return code;
}
#define FAIL(code, what) return fail(code, what, __LINE__)

// header (desc) order -> position in ProverKey::to_var_bytes / VerifierKey::to_bytes
// blob:  q_m q_l q_r q_o q_f q_c q_arith q_logic q_range fixed var s1 s2 s3 s4
// desc:  q_m q_l q_r q_o q_f q_c q_arith q_range q_logic fixed var s1 s2 s3 s4
constexpr int BLOB_POS[15] = {0, 1, 2, 3, 4, 5, 6, 8, 7, 9, 10, 11, 12, 13, 14};

int blob_check(const uint8_t* blob, uint64_t len, plonk_prover_blob_info* info) {
  if (len < 48) FAIL(PLONK_ERR_BYTES, "header");
  const uint64_t label_len = be64(blob), pk_len = be64(blob + 8), ck_len = be64(blob + 16), vk_len = be64(blob + 24);
  const uint64_t size = be64(blob + 32), constraints = be64(blob + 40);
  uint64_t need = label_len;
  if (__builtin_add_overflow(need, pk_len, &need) || __builtin_add_overflow(need, ck_len, &need) ||
      __builtin_add_overflow(need, vk_len, &need))
    FAIL(PLONK_ERR_BYTES, "section lengths overflow");
  if (len - 48 < need) FAIL(PLONK_ERR_BYTES, "sections");
  uint64_t npow = 1;
  while (npow < constraints && npow) npow <<= 1;   // checked_next_power_of_two (0 -> 1)
  if (npow == 0 || npow != size) FAIL(PLONK_ERR_DATA, "size != constraints.next_power_of_two()");
  memset(info, 0, sizeof *info);
  info->size = size;
  info->constraints = constraints;
  info->label_off = 48;
  info->label_len = label_len;

  // ---- ProverKey::from_slice
  const uint64_t pk_off = 48 + label_len;
  const uint8_t* p = blob + pk_off;
  uint64_t left = pk_len;
  auto take = [&](uint64_t k) -> const uint8_t* {
    if (left < k) return nullptr;
    const uint8_t* r = p;
    p += k;
    left -= k;
    return r;
  };
  const uint8_t* h = take(16);
  if (!h) FAIL(PLONK_ERR_BYTES, "prover key header");
  const uint64_t n = le64(h), eval_size = le64(h + 8);
  Domain d8;
  if (n > (1ull << 60) || !d8.init(n * 8)) FAIL(PLONK_ERR_DATA, "8n is not a valid evaluation domain");
  auto read_evals = [&](const uint8_t** scalars) -> int {   // Evaluations::from_slice + domain equality
    const uint8_t* e = take(eval_size);
    if (!e) return PLONK_ERR_BYTES;
    if (eval_size < DOMAIN_BYTES) return PLONK_ERR_DATA;
    if (!d8.matches(e)) return PLONK_ERR_DATA;
    if (eval_size - DOMAIN_BYTES != d8.size * SCALAR) return PLONK_ERR_DATA;
    if (!scalars_canonical(e + DOMAIN_BYTES, d8.size)) return PLONK_ERR_DATA;
    if (scalars) *scalars = e + DOMAIN_BYTES;
    return PLONK_OK;
  };
  uint64_t poly_off[15], poly_len[15];
  for (int k = 0; k < 15; ++k) {
    const uint8_t* l = take(8);
    if (!l) FAIL(PLONK_ERR_BYTES, "polynomial length");
    const uint64_t plen = le64(l);
    if (plen > n) FAIL(PLONK_ERR_DATA, "polynomial longer than n");
    const uint8_t* c = take(plen * SCALAR);
    if (!c) FAIL(PLONK_ERR_BYTES, "polynomial coefficients");
    if (!scalars_canonical(c, plen)) FAIL(PLONK_ERR_DATA, "non-canonical coefficient");
    poly_off[k] = (uint64_t)(c - blob);
    poly_len[k] = plen;
    const int rc = read_evals(nullptr);
    if (rc) FAIL(rc, "key evaluations");
  }
  for (int k = 0; k < 15; ++k) {
    info->poly_off[k] = poly_off[BLOB_POS[k]];
    info->poly_len[k] = poly_len[BLOB_POS[k]];
  }
  {   // permutation.linear_evaluations must be X over the coset (matches_linear_poly_over_coset)
    const uint8_t* s = nullptr;
    const int rc = read_evals(&s);
    if (rc) FAIL(rc, "linear evaluations");
    Fr expect = fr_generator();
    for (uint64_t i = 0; i < d8.size; ++i) {
      Fr got;
      scalar_from_bytes(s + SCALAR * i, &got);
      if (got != expect) FAIL(PLONK_ERR_DATA, "linear evaluations are not the coset points");
      expect = expect * d8.group_gen;
    }
  }
  {   // v_h_coset_8n must be X^n - 1 over the coset (matches_vanishing_poly_over_coset)
    const uint8_t* s = nullptr;
    const int rc = read_evals(&s);
    if (rc) FAIL(rc, "vanishing evaluations");
    if (n >= d8.size) FAIL(PLONK_ERR_DATA, "vanishing degree");
    Fr point = fr_generator().pow_u64(n);
    const Fr step = d8.group_gen.pow_u64(n), one = Fr::one();
    for (uint64_t i = 0; i < d8.size; ++i) {
      Fr got;
      scalar_from_bytes(s + SCALAR * i, &got);
      if (got != point - one) FAIL(PLONK_ERR_DATA, "v_h_coset_8n is not X^n - 1 over the coset");
      point = point * step;
    }
  }
  if (n != size) FAIL(PLONK_ERR_DATA, "prover_key.n != size");

  // ---- CommitKey::from_raw_var_bytes (structure + curve equation; the subgroup check runs on the GPU)
  const uint64_t ck_off = pk_off + pk_len;
  if (ck_len < 8) FAIL(PLONK_ERR_BYTES, "commit key header");
  const uint64_t npts = le64(blob + ck_off);
  if (npts == 0) FAIL(PLONK_ERR_DATA, "empty commit key");
  if (npts > (1ull << 40) || ck_len != 8 + npts * RAW_POINT) FAIL(PLONK_ERR_BYTES, "commit key length");
  {
    const Fp four = Fp::from_u64(4);
    for (uint64_t i = 0; i < npts; ++i) {
      const uint8_t* r = blob + ck_off + 8 + RAW_POINT * i;
      if (r[96] != 0) FAIL(PLONK_ERR_POINT, "identity in the commit key");
      Fp x, y;
      memcpy(x.l, r, 48);
      memcpy(y.l, r + 48, 48);
      bool in_range = true;   // limbs must be reduced for the curve test to mean anything
      for (const Fp* v : {&x, &y}) {
        bool lt = false;
        for (int k = 11; k >= 0; --k)
          if (v->l[k] != FpP::MOD[k]) { lt = v->l[k] < FpP::MOD[k]; break; }
        in_range = in_range && lt;
      }
      if (!in_range || y.sqr() != x.sqr() * x + four) FAIL(PLONK_ERR_POINT, "point not on the curve");
    }
  }
  info->srs_off = ck_off + 8;
  info->srs_points = npts;

  // ---- VerifierKey: u64 n + 15 compressed commitments inside 968 bytes
  if (vk_len < VK_BYTES) FAIL(PLONK_ERR_DATA, "verifier key length");
  // (the reference seeds the transcript with verifier_key.n AND constraints, widget.rs:218-258 /
  // transcript.rs:131-145; every compiled circuit has them equal and prover_build seeds both
  // from `constraints`, so a blob where they differ is refused rather than proved differently)
  if (le64(blob + ck_off + ck_len) != constraints) FAIL(PLONK_ERR_DATA, "verifier_key.n != constraints");
  info->vk_off = ck_off + ck_len + 8;
  // Commitment::from_reader x 15 (widget.rs:113-134): G1Affine::from_bytes refuses a non-canonical,
  // off-curve or out-of-subgroup encoding with dusk_bytes::Error::InvalidData
  for (int k = 0; k < 15; ++k)
    if (!g1_compressed_valid(blob + info->vk_off + 48 * k)) FAIL(PLONK_ERR_DATA, "verifier key commitment is not a valid compressed G1 point");
  return PLONK_OK;
}

// PublicParameters files (srs.rs:103-178): decode + structural validation, see include/plonk_hip.h
constexpr uint64_t OPENING_KEY_BYTES = 48 + 96 + 96, ADDED_BLINDING_DEGREE = 6, COMPRESSED_POINT = 48;   // key.rs:436-452, srs.rs:54

int public_parameters_check(const uint8_t* bytes, uint64_t len, uint64_t truncated_degree, int mode,
                            plonk_public_parameters_info* info) {
  memset(info, 0, sizeof *info);
  if (mode != PLONK_PP_RAW_UNCHECKED && mode != PLONK_PP_RAW && mode != PLONK_PP_COMPRESSED) FAIL(PLONK_ERR_ARG, "unknown public-parameters mode");
  if (len <= OPENING_KEY_BYTES) FAIL(PLONK_ERR_BYTES, "public parameters shorter than an opening key");   // srs.rs:165-167
  // OpeningKey::from_bytes (key.rs:596-615) -> OpeningKey::try_new (key.rs:617-648): three compressed points, each on its
  // curve, torsion-free AND not the identity — a degenerate opening key makes the pairing check trivially satisfiable.
  // (G1Affine / G2Affine::from_bytes alone accept the 0xC0 encoding; verifier-key commitments above keep accepting it.)
  if ((bytes[0] & 0x40) || (bytes[48] & 0x40) || (bytes[144] & 0x40)) FAIL(PLONK_ERR_DATA, "opening key: g, h and x_h must not be the identity");
  if (!g1_compressed_valid(bytes)) FAIL(PLONK_ERR_DATA, "opening key: g is not a valid compressed G1 point");
  // h and x_h: G2Affine::from_bytes in full (flags, canonical coordinates, on the twist curve, order q) — hostg2.hpp
  if (!g2_compressed_valid(bytes + 48)) FAIL(PLONK_ERR_DATA, "opening key: h is not a valid compressed G2 point");
  if (!g2_compressed_valid(bytes + 144)) FAIL(PLONK_ERR_DATA, "opening key: x_h is not a valid compressed G2 point");
  info->opening_key_off = 0;
  const uint8_t* ck = bytes + OPENING_KEY_BYTES;
  const uint64_t ck_len = len - OPENING_KEY_BYTES;
  uint64_t npts;
  if (mode == PLONK_PP_COMPRESSED) {   // CommitKey::from_slice (key.rs:319-326): chunks(48).map(G1Affine::from_slice)
    if (ck_len % COMPRESSED_POINT) FAIL(PLONK_ERR_DATA, "commit key: short last chunk (dusk_bytes BadLength)");
    npts = ck_len / COMPRESSED_POINT;
    info->points_off = OPENING_KEY_BYTES;
    info->point_stride = COMPRESSED_POINT;
  } else {
    if (ck_len < 8) FAIL(PLONK_ERR_BYTES, "commit key header");
    const uint64_t count = le64(ck);
    if (mode == PLONK_PP_RAW) {   // CommitKey::from_raw_var_bytes (key.rs:263-300)
      if (count == 0) FAIL(PLONK_ERR_DATA, "empty commit key");
      if (count > (1ull << 40) || ck_len != 8 + count * RAW_POINT) FAIL(PLONK_ERR_BYTES, "commit key length");
      npts = count;
    } else {                      // CommitKey::from_slice_unchecked (key.rs:243-258): chunks_exact(97).zip(0..count)
      const uint64_t chunks = (ck_len - 8) / RAW_POINT;
      npts = count < chunks ? count : chunks;
      if (npts == 0) FAIL(PLONK_ERR_BYTES, "commit key holds no point");
    }
    info->points_off = OPENING_KEY_BYTES + 8;
    info->point_stride = RAW_POINT;
  }
  info->points_total = npts;
  uint64_t keep = npts;
  if (truncated_degree) {   // PublicParameters::trim -> CommitKey::truncate (srs.rs:188-196, key.rs:336-355)
    uint64_t d;
    if (__builtin_add_overflow(truncated_degree, ADDED_BLINDING_DEGREE, &d) || d > npts - 1) FAIL(PLONK_ERR_DEGREE, "TruncatedDegreeTooLarge");
    if (d == 1) d = 2;      // (unreachable with the +6, kept for the literal rule)
    keep = d + 1;
  }
  info->points_kept = keep;
  // The validating decoders look at EVERY point of the file before trim() drops the tail (key.rs:263-300, :319-326): a bad
  // point beyond the kept prefix refuses the file there, so it does here.  from_slice_unchecked looks at none; the kept
  // prefix is still scanned for identities (which no table row can hold — see plonk_hip.h, a documented divergence).
  const uint64_t scan = mode == PLONK_PP_RAW_UNCHECKED ? keep : npts;
  for (uint64_t i = 0; i < scan; ++i) {
    const uint8_t* r = bytes + info->points_off + info->point_stride * i;
    if (mode == PLONK_PP_COMPRESSED) {   // the flag and range half of G1Affine::from_bytes; the square root is the GPU's
      if (!(r[0] & 0x80)) FAIL(PLONK_ERR_DATA, "commit key point without the compression flag");
      if (r[0] & 0x40) {
        bool zero = !(r[0] & 0x3f);
        for (int k = 1; k < 48; ++k) zero = zero && r[k] == 0;
        if (!zero) FAIL(PLONK_ERR_DATA, "malformed identity encoding");
        FAIL(PLONK_ERR_POINT, "identity in the commit key");
      }
      continue;
    }
    if (r[96] != 0) FAIL(PLONK_ERR_POINT, "identity in the commit key");
    if (mode == PLONK_PP_RAW) {   // reduced limbs, so that the curve test on the GPU means what it says
      for (int c = 0; c < 2; ++c) {
        bool lt = false;
        for (int k = 11; k >= 0; --k) {
          const uint32_t w = le32(r + 48 * c + 4 * k);
          if (w != FpP::MOD[k]) { lt = w < FpP::MOD[k]; break; }
        }
        if (!lt) FAIL(PLONK_ERR_POINT, "coordinate not reduced");
      }
    }
  }
  return PLONK_OK;
}

}  // namespace
}  // namespace plonk
