// Prover::try_from_bytes for the device prover: turns the blob a dusk-plonk `Prover::to_bytes()`
// writes (src/compiler/prover.rs:238-263) into a device-resident prover, so a compiled circuit
// can be handed over without any Rust-side glue.
//
// Decoding mirrors, check for check:
//   Prover::try_from_bytes            src/compiler/prover.rs:266-345
//   ProverKey::from_slice             src/proof_system/widget.rs:449-625
//   Evaluations::from_slice           src/fft/evaluations.rs:64-90
//   EvaluationDomain::{from_bytes,new,matches_*}   src/fft/domain.rs:81-105,122-158,307-352
//   Polynomial::from_slice            src/fft/polynomial.rs:152-163
//   CommitKey::from_raw_var_bytes     src/commitment_scheme/kzg10/key.rs:263-300
//   VerifierKey (layout only)         src/proof_system/widget.rs:84-134
// Error mapping: Error::NotEnoughBytes -> PLONK_ERR_BYTES, dusk_bytes::Error::InvalidData (and
// the non-canonical-scalar decode failures) -> PLONK_ERR_DATA, Error::PointMalformed ->
// PLONK_ERR_POINT.
//
// What is NOT taken from the blob: the 15 x 8n coset evaluation arrays.  The reference trusts
// them; this backend rebuilds them on the device from the coefficient forms (prover_build), so
// they are only validated structurally (domain header, length, canonical scalars).  For a blob
// written by the reference both are the same numbers.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/plonk_hip.h"
#include "plonk_internal.hpp"
#include "hostg2.hpp"
#include "g1codec.cuh"

namespace plonk {
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

// one lane per 48-byte chunk: G1Affine::from_bytes minus the subgroup test (srs_validate_kernel's job); flag |= 1 invalid, 2 identity
__global__ void __launch_bounds__(64) g1_decompress_kernel(const uint8_t* __restrict__ in, uint64_t n, G1Affine* __restrict__ out,
                                                           int* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine a;
  const int rc = g1_decompress48(in + COMPRESSED_POINT * i, &a);
  if (rc != G1DEC_OK) { atomicOr(flag, rc == G1DEC_IDENTITY ? 2 : 1); return; }
  out[i] = a;
}

}  // namespace
}  // namespace plonk

using namespace plonk;

extern "C" {

int plonk_prover_blob_check(const uint8_t* blob, uint64_t len, plonk_prover_blob_info* info) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!blob || !info) return (set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  return blob_check(blob, len, info);
  });
}

int plonk_public_parameters_check(const uint8_t* bytes, uint64_t len, uint64_t truncated_degree, int mode,
                                  plonk_public_parameters_info* info) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!bytes || !info) return (set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  return public_parameters_check(bytes, len, truncated_degree, mode, info);
  });
}

int plonk_srs_load_public_parameters(plonk_ctx* ctx, const uint8_t* bytes, uint64_t len, uint64_t truncated_degree,
                                     int mode, uint8_t opening_key_out[240], uint64_t* points_loaded) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx || !bytes) return (set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  plonk_public_parameters_info info;
  int rc = public_parameters_check(bytes, len, truncated_degree, mode, &info);
  if (rc) return rc;
  if (mode == PLONK_PP_COMPRESSED) {
    // 48-byte chunks -> raw affine points on the device (a square root each), subgroup test, window tables: the decoded
    // key never visits the host
    Ctx& c = ctx->c;
    CTX_ENTER(c, api_fn);
    HIP_TRY(hipSetDevice(c.device));
    uint8_t* in = nullptr;
    G1Affine* pts = nullptr;
    int* flag = nullptr;
    const uint64_t n = info.points_total;   // every chunk is decoded and tested, the kept prefix is loaded
    hipError_t e = hipMalloc((void**)&in, COMPRESSED_POINT * n);
    if (e == hipSuccess) e = hipMalloc((void**)&pts, sizeof(G1Affine) * n);
    if (e == hipSuccess) e = hipMalloc((void**)&flag, 2 * sizeof(int));
    int bad[2] = {0, 0};
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, 2 * sizeof(int), c.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(in, bytes + info.points_off, COMPRESSED_POINT * n, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(g1_decompress_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, in, n, pts, flag);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad[0], flag, sizeof(int), hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    rc = e == hipSuccess ? PLONK_OK : PLONK_ERR_HIP;
    if (rc == PLONK_OK && bad[0]) {   // (an identity was already refused by the host pass)
      set_last_error("InvalidData", "commit key chunk is not a valid compressed G1 point", __FILE__, __LINE__);
      rc = PLONK_ERR_DATA;
    }
    if (rc == PLONK_OK) rc = srs_validate_device(&c, pts, n, flag + 1);   // is_torsion_free
    if (rc == PLONK_OK && hipMemcpyAsync(&bad[1], flag + 1, sizeof(int), hipMemcpyDeviceToHost, c.stream) != hipSuccess) rc = PLONK_ERR_HIP;
    if (rc == PLONK_OK && hipStreamSynchronize(c.stream) != hipSuccess) rc = PLONK_ERR_HIP;
    if (rc == PLONK_OK && bad[1]) {
      set_last_error("InvalidData", "commit key point outside the prime-order subgroup", __FILE__, __LINE__);
      rc = PLONK_ERR_DATA;
    }
    if (rc == PLONK_OK) rc = srs_load_device(&c, pts, info.points_kept);
    (void)hipStreamSynchronize(c.stream);
    (void)hipFree(in);
    (void)hipFree(pts);
    (void)hipFree(flag);
    if (e != hipSuccess) set_last_error("plonk_srs_load_public_parameters", hipGetErrorString(e), __FILE__, __LINE__);
    if (rc) return rc;
  } else {
    // 97-byte raw points -> x || y.  PLONK_PP_RAW: is_on_curve & is_torsion_free for every point OF THE FILE, on the GPU
    // (key.rs:287-293), then the trimmed prefix is loaded; unchecked: the prefix only, the rest is never touched.
    const uint64_t nval = mode == PLONK_PP_RAW ? info.points_total : info.points_kept;
    std::vector<uint8_t> xy((size_t)nval * 96);
    for (uint64_t i = 0; i < nval; ++i) memcpy(&xy[96 * i], bytes + info.points_off + RAW_POINT * i, 96);
    if (mode == PLONK_PP_RAW) {
      rc = plonk_srs_validate(ctx, xy.data(), nval);
      if (rc) return rc;
    }
    rc = plonk_srs_load(ctx, xy.data(), info.points_kept);
    if (rc) return rc;
  }
  if (opening_key_out) memcpy(opening_key_out, bytes + info.opening_key_off, OPENING_KEY_BYTES);
  if (points_loaded) *points_loaded = info.points_kept;
  return PLONK_OK;
  });
}

int plonk_prover_from_bytes(plonk_ctx* ctx, const uint8_t* blob, uint64_t len, plonk_prover** out) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx || !blob || !out) return (set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  plonk_prover_blob_info info;
  int rc = blob_check(blob, len, &info);
  if (rc) return rc;
  // commit key: 97-byte raw points -> x || y, subgroup check + window tables on the device
  std::vector<uint8_t> xy((size_t)info.srs_points * 96);
  for (uint64_t i = 0; i < info.srs_points; ++i) memcpy(&xy[96 * i], blob + info.srs_off + RAW_POINT * i, 96);
  rc = plonk_srs_validate(ctx, xy.data(), info.srs_points);   // every point, as the reference does
  if (rc) return rc;
  // prove() commits to at most size + 7 coefficients: only that prefix of the key gets window tables (a key trimmed
  // by Compiler::compile has up to 2 * size + 6 points, compiler.rs:121-124 — twice the table memory for nothing)
  const uint64_t used = info.srs_points < info.size + 8 ? info.srs_points : info.size + 8;
  rc = plonk_srs_load(ctx, xy.data(), used);
  if (rc) return rc;
  std::vector<uint8_t>().swap(xy);
  // key polynomials: canonical little-endian scalars -> Montgomery limbs
  std::vector<std::vector<Fr>> polys(15);
  plonk_prover_desc desc;
  memset(&desc, 0, sizeof desc);
  desc.constraints = info.constraints;
  desc.label = blob + info.label_off;
  desc.label_len = info.label_len;
  for (int k = 0; k < 15; ++k) {
    polys[k].resize(info.poly_len[k] ? info.poly_len[k] : 1);
    for (uint64_t i = 0; i < info.poly_len[k]; ++i) scalar_from_bytes(blob + info.poly_off[k] + SCALAR * i, &polys[k][i]);
    uint64_t plen = info.poly_len[k];
    while (plen && polys[k][plen - 1].is_zero()) --plen;   // Polynomial::from_slice truncates leading zeros
    desc.polys[k] = reinterpret_cast<const uint64_t*>(polys[k].data());
    desc.poly_len[k] = plen;
  }
  uint8_t vk[15 * 48];
  for (int k = 0; k < 15; ++k) memcpy(vk + 48 * k, blob + info.vk_off + 48 * BLOB_POS[k], 48);
  desc.vk_commitments = vk;
  return plonk_prover_create(ctx, &desc, out);
  });
}

}  // extern "C"
