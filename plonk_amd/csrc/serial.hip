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

#include "serial_check.hpp"   // blob_check, public_parameters_check and their helpers (HIP-free, fuzzed on the host)

namespace plonk {
namespace {

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
