/* Plain C against include/plonk_hip.h: a serialized dusk-plonk Prover (Prover::to_bytes(), reference
 * src/compiler/prover.rs:238-263) + the four padded wire columns + the 14 blinding scalars -> the 1008 Proof bytes.
 * This is the whole integration surface a non-Rust service needs; no Python, no torch.
 *
 *   cc -std=c99 -Iinclude examples/prove_from_blob.c -Lplonk_amd/lib -lplonk_hip -Wl,-rpath,$PWD/plonk_amd/lib -o prove_from_blob
 *   ./prove_from_blob circuit.prover wires.bin blinders.bin proof.bin [public_inputs.bin]
 *
 * wires.bin         4 x size BlsScalar values (a, b, c, d columns; size = the prover's domain size), 32-byte Montgomery
 *                   limbs exactly as BlsScalar.0 lies in memory
 * blinders.bin      14 x 32 bytes, the scalars Prover::prove draws from its RNG, in draw order (prover.rs:444-587)
 * public_inputs.bin optional: count x (u64 LE row index, 32-byte value)
 * Exit code: 0 ok, 1 library / device error (message on stderr), 2 usage. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "plonk_hip.h"

static unsigned char* slurp(const char* path, uint64_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); return NULL; }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char* buf = (unsigned char*)malloc(n > 0 ? (size_t)n : 1);
  if (buf && n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
  fclose(f);
  *len = (uint64_t)n;
  return buf;
}

static int fail(const char* what, int rc) {
  fprintf(stderr, "%s failed: code %d (%s)\n", what, rc, plonk_last_error() ? plonk_last_error() : "");
  return 1;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s prover.blob wires.bin blinders.bin proof.out [public_inputs.bin]\n", argv[0]);
    return 2;
  }
  uint64_t blob_len = 0, wires_len = 0, bl_len = 0, pi_len = 0;
  unsigned char* blob = slurp(argv[1], &blob_len);
  unsigned char* wires = slurp(argv[2], &wires_len);
  unsigned char* bl = slurp(argv[3], &bl_len);
  unsigned char* pi = argc > 5 ? slurp(argv[5], &pi_len) : NULL;
  if (!blob || !wires || !bl || (argc > 5 && !pi)) return 2;

  /* host-side decode first: a malformed blob is refused without touching a device */
  plonk_prover_blob_info info;
  int rc = plonk_prover_blob_check(blob, blob_len, &info);
  if (rc) return fail("plonk_prover_blob_check", rc);
  if (wires_len != 4 * 32 * info.size || bl_len != 14 * 32 || pi_len % 40) {
    fprintf(stderr, "expected 4 x %llu x 32 wire bytes, 14 x 32 blinder bytes, public inputs in 40-byte records\n",
            (unsigned long long)info.size);
    return 2;
  }
  const uint64_t pi_count = pi_len / 40;
  uint64_t* pi_idx = (uint64_t*)malloc(8 * (pi_count ? pi_count : 1));
  uint64_t* pi_val = (uint64_t*)malloc(32 * (pi_count ? pi_count : 1));
  for (uint64_t i = 0; i < pi_count; ++i) {
    memcpy(&pi_idx[i], pi + 40 * i, 8);
    memcpy(&pi_val[4 * i], pi + 40 * i + 8, 32);
  }

  plonk_ctx* ctx = NULL;
  const int device = 0;
  rc = plonk_ctx_create(&ctx, &device, 1);
  if (rc) return fail("plonk_ctx_create", rc);                       /* PLONK_ERR_NO_GPU without an MI355X: no CPU fallback */
  plonk_prover* prover = NULL;
  rc = plonk_prover_from_bytes(ctx, blob, blob_len, &prover);        /* Prover::try_from_bytes + commit key onto the device */
  if (rc) { plonk_ctx_destroy(ctx); return fail("plonk_prover_from_bytes", rc); }
  const uint64_t* cols[4];
  for (int k = 0; k < 4; ++k) cols[k] = (const uint64_t*)(wires + (size_t)k * 32 * info.size);
  uint8_t proof[1008];
  rc = plonk_prover_prove(prover, cols, pi_idx, pi_val, pi_count, (const uint64_t*)bl, proof);
  if (rc) { plonk_prover_destroy(prover); plonk_ctx_destroy(ctx); return fail("plonk_prover_prove", rc); }
  FILE* out = fopen(argv[4], "wb");
  if (!out || fwrite(proof, 1, sizeof proof, out) != sizeof proof) { fprintf(stderr, "cannot write %s\n", argv[4]); return 1; }
  fclose(out);
  printf("proof: %u bytes for a %llu-gate domain (%llu constraints) -> %s\n", (unsigned)sizeof proof,
         (unsigned long long)info.size, (unsigned long long)info.constraints, argv[4]);
  plonk_prover_destroy(prover);
  plonk_ctx_destroy(ctx);
  free(blob); free(wires); free(bl); free(pi); free(pi_idx); free(pi_val);
  return 0;
}
