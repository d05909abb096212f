// libFuzzer harness over the two byte parsers of the C-ABI that read UNTRUSTED input and cannot be pinned to
// reference-produced files in this environment:
//   plonk_prover_blob_check          -> blob_check               (Prover::try_from_bytes, prover.rs:266-345)
//   plonk_public_parameters_check    -> public_parameters_check  (PublicParameters::from_slice[_unchecked] + trim, srs.rs:103-196)
// It compiles EXACTLY the product's code (plonk_amd/csrc/serial_check.hpp, HIP-free) for the host under AddressSanitizer +
// UndefinedBehaviorSanitizer.  After a successful check the harness reads every byte range the returned info structure
// names, the way plonk_prover_from_bytes / plonk_srs_load_public_parameters go on to — an offset that slipped through
// validation is an ASan report here, not silent garbage on the device.  Test infrastructure; never linked into the product.
//
//   clang++ -std=c++17 -O1 -g -fsanitize=fuzzer,address,undefined -fno-sanitize-recover=undefined \
//           tests/fuzz/fuzz_serial.cpp -o tests/fuzz/_build/fuzz_serial
//   tests/fuzz/_build/fuzz_serial -max_total_time=300 <work corpus> tests/fuzz/corpus        (tests/fuzz/run.sh)
//
// Input: byte 0 selects the parser (0: prover blob; 1 / 2 / 3: public parameters raw-unchecked / raw / compressed), bytes 1-2 the
// truncated degree of a public-parameters file (little-endian; 0 = no trim); the rest is the file.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../plonk_amd/csrc/serial_check.hpp"

namespace plonk {
void set_last_error(const char*, const char*, const char*, int) {}
}  // namespace plonk

static volatile uint8_t g_sink;
static void touch(const uint8_t* base, uint64_t len, uint64_t off, uint64_t cnt) {
  if (cnt == 0) return;
  if (off > len || cnt > len - off) {   // the info structure points outside the input: a validation hole
    fprintf(stderr, "info range [%llu, +%llu) outside the %llu input bytes\n", (unsigned long long)off, (unsigned long long)cnt, (unsigned long long)len);
    abort();
  }
  uint8_t acc = 0;
  for (uint64_t i = 0; i < cnt; ++i) acc ^= base[off + i];
  g_sink = acc;
}

extern "C" int LLVMFuzzerTestOneInput(const uint8_t* data, size_t size) {
  if (size < 3) return 0;
  const int which = data[0] & 3;
  const uint64_t degree = (uint64_t)data[1] | ((uint64_t)data[2] << 8);
  // the parser gets an exact-size heap copy: one byte read past the end is a heap-buffer-overflow
  const uint64_t len = size - 3;
  uint8_t* buf = (uint8_t*)malloc(len ? len : 1);
  memcpy(buf, data + 3, len);
  if (which == 0) {
    plonk_prover_blob_info info;
    const int rc = plonk::blob_check(buf, len, &info);
    if (rc == PLONK_OK) {
      touch(buf, len, info.label_off, info.label_len);
      for (int k = 0; k < 15; ++k) touch(buf, len, info.poly_off[k], info.poly_len[k] * 32);
      touch(buf, len, info.srs_off, info.srs_points * 97);
      touch(buf, len, info.vk_off, 15 * 48);
      if (info.size == 0 || (info.size & (info.size - 1)) || info.constraints > info.size) abort();
      for (int k = 0; k < 15; ++k) if (info.poly_len[k] > info.size) abort();
    } else if (rc != PLONK_ERR_BYTES && rc != PLONK_ERR_DATA && rc != PLONK_ERR_POINT) {
      abort();   // only the reference's three decode errors may come back
    }
  } else {
    plonk_public_parameters_info info;
    const int mode = which == 1 ? PLONK_PP_RAW_UNCHECKED : which == 2 ? PLONK_PP_RAW : PLONK_PP_COMPRESSED;
    const int rc = plonk::public_parameters_check(buf, len, degree, mode, &info);
    if (rc == PLONK_OK) {
      touch(buf, len, info.opening_key_off, 240);
      if (info.points_kept > info.points_total || info.points_kept == 0) abort();
      touch(buf, len, info.points_off, info.points_total * info.point_stride);
    } else if (rc != PLONK_ERR_BYTES && rc != PLONK_ERR_DATA && rc != PLONK_ERR_POINT && rc != PLONK_ERR_DEGREE) {
      abort();
    }
  }
  free(buf);
  return 0;
}
