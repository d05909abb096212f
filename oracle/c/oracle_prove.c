/* CPU restatement (plain C + OpenMP) of the WHOLE Prover::prove — TEST INFRASTRUCTURE ONLY.
 *
 * This translation unit includes oracle.c (fields, best_fft, Pippenger) and adds every other
 * step of prove_inner so that the GPU prover can be compared byte for byte at the BASELINE
 * sizes (2^12 .. 2^20 gates), where the big-int Python oracle is too slow; it is also the
 * `cpu_baseline` leg of bench.py ("port": the reference's algorithm, 64-bit limbs, OpenMP in
 * place of rayon).  The product (plonk_amd/) never links or calls this file.
 *
 * Restates (reference file:line):
 *   Transcript (merlin 3.0: STROBE-128 / Keccak-f[1600]) + TranscriptProtocol   src/transcript.rs:90-145
 *   VerifierKey::seed_transcript                                                 src/proof_system/widget.rs:218-258
 *   Prover::new cached state (8n coset evaluations, sigma evaluations,
 *     vanishing inverses)                                                        src/compiler/prover.rs:53-115, src/compiler.rs:310-425
 *   prove_inner rounds 1-5                                                       src/compiler/prover.rs:415-761
 *   blind_poly / permutation vector                                              prover.rs:139-152, src/composer/permutation.rs:213-294
 *   quotient_poly::compute (8n coset, every widget)                              src/proof_system/quotient_poly.rs:20-310,
 *                                                                                widget/{arithmetic,range,logic,ecc/..,permutation}/proverkey.rs
 *   linearization_poly::compute                                                  src/proof_system/linearization_poly.rs:168-264
 *   compute_aggregate_witness / ruffini                                          src/commitment_scheme/kzg10/key.rs:394-417, src/fft/polynomial.rs:345-367
 *   Proof::to_bytes                                                              src/proof_system/proof.rs:137-162
 * Pinning: reproduces the reference KAT digest (prover.rs:1151-1158) and equals the big-int
 * oracle byte for byte on the widget / public-input circuits (tests/test_oracle_c_prove.py).
 */
#include "oracle.c"

#include <stdio.h>

/* ------------------------------------------------------------------ small Fr helpers */
static const u64 FR_R3[4] = {0xc62c1807439b73afull, 0x1b3e0d188cf06990ull, 0x73d13c71c7b5f418ull, 0x6e2a5bb9c8db33e9ull};
static inline fr fr_one(void) { fr r; memcpy(r.l, FR_ONE, 32); return r; }
static inline fr fr_zero(void) { fr r; memset(r.l, 0, 32); return r; }
static inline int fr_is_zero(fr a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline int fr_eq(fr a, fr b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
static inline fr fr_neg(fr a) { return fr_sub(fr_zero(), a); }
static inline fr fr_sqr(fr a) { return fr_mul(a, a); }
static inline fr fr_dbl(fr a) { return fr_add(a, a); }
static inline fr fr_small(u64 v) { return fr_from_u64(v); }
static inline fr fr_ld(const u64* p) { fr r; memcpy(r.l, p, 32); return r; }
/* BlsScalar::to_bytes: canonical 32-byte little-endian */
static void fr_to_bytes(fr a, uint8_t out[32]) {
  fr one_raw = {{1, 0, 0, 0}};
  fr c = fr_mul(a, one_raw);
  memcpy(out, c.l, 32);
}
/* BlsScalar::from_bytes_wide: 512-bit little-endian integer mod q (transcript.rs:98-103) */
static fr fr_from_bytes_wide(const uint8_t b[64]) {
  fr lo, hi, r2, r3;
  memcpy(lo.l, b, 32); memcpy(hi.l, b + 32, 32);
  memcpy(r2.l, FR_R2, 32); memcpy(r3.l, FR_R3, 32);
  return fr_add(fr_mul(lo, r2), fr_mul(hi, r3));   /* lo R + hi 2^256 R */
}

/* ------------------------------------------------------------------ Keccak-f[1600] / STROBE-128 / Merlin */
static const u64 KRC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static inline u64 rol64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static void keccak_f(uint8_t st8[200]) {
  static const int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  u64 st[25], bc[5];
  memcpy(st, st8, 200);
  for (int r = 0; r < 24; ++r) {
    for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; ++i) { u64 t = bc[(i + 4) % 5] ^ rol64(bc[(i + 1) % 5], 1); for (int j = 0; j < 25; j += 5) st[j + i] ^= t; }
    u64 t = st[1];
    for (int i = 0; i < 24; ++i) { int j = piln[i]; u64 b = st[j]; st[j] = rol64(t, rotc[i]); t = b; }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
      for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KRC[r];
  }
  memcpy(st8, st, 200);
}
#define STROBE_R 166
typedef struct { uint8_t st[200]; int pos, pos_begin, cur_flags; } strobe;
static void strobe_run_f(strobe* s) {
  s->st[s->pos] ^= (uint8_t)s->pos_begin;
  s->st[s->pos + 1] ^= 0x04;
  s->st[STROBE_R + 1] ^= 0x80;
  keccak_f(s->st);
  s->pos = 0; s->pos_begin = 0;
}
static void strobe_absorb(strobe* s, const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) { s->st[s->pos++] ^= d[i]; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void strobe_begin(strobe* s, int flags, int more) {
  if (more) return;
  uint8_t hdr[2] = {(uint8_t)s->pos_begin, (uint8_t)flags};
  s->pos_begin = s->pos + 1;
  s->cur_flags = flags;
  strobe_absorb(s, hdr, 2);
  if ((flags & (4 | 32)) && s->pos != 0) strobe_run_f(s);
}
static void strobe_meta_ad(strobe* s, const void* d, size_t n, int more) { strobe_begin(s, 16 | 2, more); strobe_absorb(s, (const uint8_t*)d, n); }
static void strobe_ad(strobe* s, const void* d, size_t n, int more) { strobe_begin(s, 2, more); strobe_absorb(s, (const uint8_t*)d, n); }
static void strobe_prf(strobe* s, uint8_t* out, size_t n) {
  strobe_begin(s, 1 | 2 | 4, 0);
  for (size_t i = 0; i < n; ++i) { out[i] = s->st[s->pos]; s->st[s->pos++] = 0; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void tr_append(strobe* s, const char* label, const void* msg, uint32_t len) {
  strobe_meta_ad(s, label, strlen(label), 0);
  strobe_meta_ad(s, &len, 4, 1);            /* u32 little-endian (host is LE) */
  strobe_ad(s, msg, len, 0);
}
static void tr_init(strobe* s, const uint8_t* label, size_t len) {
  memset(s, 0, sizeof *s);
  const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
  memcpy(s->st, hdr, 6);
  memcpy(s->st + 6, "STROBEv1.0.2", 12);
  keccak_f(s->st);
  strobe_meta_ad(s, "Merlin v1.0", 11, 0);
  strobe_meta_ad(s, "dom-sep", 7, 0);
  uint32_t l32 = (uint32_t)len;
  strobe_meta_ad(s, &l32, 4, 1);
  strobe_ad(s, label, len, 0);
}
static void tr_scalar(strobe* s, const char* label, fr v) { uint8_t b[32]; fr_to_bytes(v, b); tr_append(s, label, b, 32); }
static fr tr_challenge(strobe* s, const char* label) {
  uint8_t b[64]; uint32_t n = 64;
  strobe_meta_ad(s, label, strlen(label), 0);
  strobe_meta_ad(s, &n, 4, 1);
  strobe_prf(s, b, 64);
  return fr_from_bytes_wide(b);
}
static void tr_domain_sep(strobe* s, u64 n) { tr_append(s, "dom-sep", "circuit_size", 12); tr_append(s, "n", &n, 8); }

/* ------------------------------------------------------------------ G1 compression */
/* Commitment::to_bytes = G1Affine::to_bytes: 48-byte big-endian x, flags 0x80 | 0x40 inf | 0x20 y > -y */
static void g1_compress97(const uint8_t raw[97], uint8_t out[48]) {
  memset(out, 0, 48);
  if (raw[96]) { out[0] = 0xc0; return; }
  fp x, y, one_raw = {{1, 0, 0, 0, 0, 0}};
  memcpy(x.l, raw, 48); memcpy(y.l, raw + 48, 48);
  x = fp_mul(x, one_raw); y = fp_mul(y, one_raw);     /* canonical */
  for (int i = 0; i < 6; ++i) for (int b = 0; b < 8; ++b) out[47 - (8 * i + b)] = (uint8_t)(x.l[i] >> (8 * b));
  out[0] |= 0x80;
  /* y > p - y  <=>  y > (p - 1) / 2 */
  static const u64 HALF[6] = {0xdcff7fffffffd555ull, 0x0f55ffff58a9ffffull, 0xb39869507b587b12ull, 0xb23ba5c279c2895full, 0x258dd3db21a5d66bull, 0x0d0088f51cbff34dull};
  int gt = 0;
  for (int i = 5; i >= 0; --i) { if (y.l[i] != HALF[i]) { gt = y.l[i] > HALF[i]; break; } }
  if (gt) out[0] |= 0x20;
}

/* ------------------------------------------------------------------ prover state */
enum { K_QM, K_QL, K_QR, K_QO, K_QF, K_QC, K_QARITH, K_QRANGE, K_QLOGIC, K_QFIXED, K_QVAR, K_S1, K_S2, K_S3, K_S4, K_COUNT };
/* VerifierKey::seed_transcript order (widget.rs:229-254) */
static const int VK_ORDER[15] = {K_QM, K_QL, K_QR, K_QO, K_QC, K_QF, K_QARITH, K_QRANGE, K_QLOGIC, K_QVAR, K_QFIXED, K_S1, K_S2, K_S3, K_S4};
static const char* VK_LABEL[15] = {"q_m", "q_l", "q_r", "q_o", "q_c", "q_f", "q_arith", "q_range", "q_logic",
                                   "q_variable_group_add", "q_fixed_group_add", "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"};

typedef struct {
  u64 n, n8, constraints; uint32_t logn;
  uint8_t* label; u64 label_len;
  fr* polys[K_COUNT]; u64 poly_len[K_COUNT];   /* trimmed lengths */
  fr* ev8[K_COUNT + 1];                        /* 8n coset evaluations; [K_COUNT] = "linear" (coset of X) */
  fr* sigma_n[4];                              /* sigma evaluations over the proving domain (prover.rs:95-100) */
  fr vh[8], vinv[8];                           /* vanishing polynomial over the coset (period 8) and inverses */
  uint8_t* srs; u64 srs_n;
  uint8_t vk[15][48];                          /* POLY order (K_*) */
  int threads;
  int trapdoor; fr tau, gscalar;               /* oracle_prover_set_trapdoor: the key is [g tau^i] G with KNOWN tau, g */
  int version;                                 /* 3 (default) or 2: prove_with_version (prover.rs:365-413) */
} oprover;

static const u64 G1_GEN_X[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
static const u64 G1_GEN_Y[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};

static void j_to_affine(g1j p, g1a* out) {   /* p finite */
  fp zi = fp_inv(p.z), zi2 = fp_mul(zi, zi);
  out->x = fp_mul(p.x, zi2);
  out->y = fp_mul(p.y, fp_mul(zi2, zi));
}

static u64 trimmed_len(const fr* p, u64 len) { while (len && fr_is_zero(p[len - 1])) --len; return len; }

/* CommitKey::commit (key.rs:376-388) on a trimmed polynomial -> 48-byte commitment; -3 = PolynomialDegreeTooLarge */
static void commit_trapdoor(const oprover* P, const fr* poly, u64 len, uint8_t raw[97]);
static int commit48(const oprover* P, const fr* poly, u64 len, uint8_t out[48]) {
  len = trimmed_len(poly, len);
  if (len > P->srs_n) return -3;
  uint8_t raw[97];
  if (P->trapdoor) commit_trapdoor(P, poly, len, raw);
  else oracle_msm(P->srs, (const u64*)poly, len, raw, P->threads);
  g1_compress97(raw, out);
  return 0;
}

static fr omega_of(uint32_t logn) { fr w; memcpy(w.l, FR_ROOT, 32); for (uint32_t i = logn; i < 32; ++i) w = fr_mul(w, w); return w; }

void oracle_prover_free(oprover* P) {
  if (!P) return;
  for (int k = 0; k < K_COUNT; ++k) { free(P->polys[k]); free(P->ev8[k]); }
  free(P->ev8[K_COUNT]);
  for (int k = 0; k < 4; ++k) free(P->sigma_n[k]);
  free(P->label); free(P->srs); free(P);
}

/* polys[k]: poly_len[k] coefficients (Montgomery), K_* order; srs96: srs_n x 96 B; vk48: 15 x 48 B in K_* order or
 * NULL (Compiler::preprocess commits, compiler.rs:213-232). */
oprover* oracle_prover_new(u64 constraints, const uint8_t* label, u64 label_len, const u64* const* polys, const u64* poly_len,
                           const uint8_t* srs96, u64 srs_n, const uint8_t* vk48, int threads) {
  if (threads <= 0) threads = omp_get_max_threads();
  oprover* P = (oprover*)calloc(1, sizeof(oprover));
  P->threads = threads;
  P->constraints = constraints;
  u64 n = 1; uint32_t L = 0;
  while (n < constraints) { n <<= 1; ++L; }
  P->n = n; P->logn = L; P->n8 = 8 * n;
  P->label = (uint8_t*)malloc(label_len + 1); memcpy(P->label, label, label_len); P->label_len = label_len;
  P->srs = (uint8_t*)malloc(96 * (srs_n ? srs_n : 1)); memcpy(P->srs, srs96, 96 * srs_n); P->srs_n = srs_n;
  for (int k = 0; k < K_COUNT; ++k) {
    P->polys[k] = (fr*)calloc(n, 32);
    if (poly_len[k] > n) { oracle_prover_free(P); return NULL; }
    memcpy(P->polys[k], polys[k], 32 * poly_len[k]);
    P->poly_len[k] = trimmed_len(P->polys[k], poly_len[k]);
    P->ev8[k] = (fr*)malloc(32 * P->n8);
    memcpy(P->ev8[k], P->polys[k], 32 * n);
    oracle_ntt((u64*)P->ev8[k], L + 3, 0, 1, P->poly_len[k], threads);      /* compiler.rs:312-377 */
  }
  P->ev8[K_COUNT] = (fr*)calloc(P->n8, 32);
  P->ev8[K_COUNT][1] = fr_one();
  oracle_ntt((u64*)P->ev8[K_COUNT], L + 3, 0, 1, 2, threads);               /* linear_evaluations */
  for (int k = 0; k < 4; ++k) {
    P->sigma_n[k] = (fr*)malloc(32 * n);
    memcpy(P->sigma_n[k], P->polys[K_S1 + k], 32 * n);
    oracle_ntt((u64*)P->sigma_n[k], L, 0, 0, n, threads);
  }
  {  /* vanishing_poly_over_coset (domain.rs:338-351) — period 8 — and its inverses (prover.rs:78-91) */
    fr g; memcpy(g.l, FR_GEN, 32);
    fr point = fr_pow(g, n);
    const fr step = fr_pow(omega_of(L + 3), n);
    for (int i = 0; i < 8; ++i) { P->vh[i] = fr_sub(point, fr_one()); P->vinv[i] = fr_inv(P->vh[i]); point = fr_mul(point, step); }
  }
  if (vk48) memcpy(P->vk, vk48, 15 * 48);
  else for (int k = 0; k < K_COUNT; ++k) if (commit48(P, P->polys[k], P->poly_len[k], P->vk[k])) { oracle_prover_free(P); return NULL; }
  return P;
}
void oracle_prover_vk(const oprover* P, uint8_t out[15 * 48]) { memcpy(out, P->vk, 15 * 48); }

/* Test keys are synthetic: point i is [g tau^i] G1 for KNOWN tau and g (oracle_srs_generate).  With the trapdoor the
 * commitment sum_i p_i [g tau^i] G is the single point [g p(tau)] G — the SAME group element CommitKey::commit
 * (key.rs:376-388) returns, obtained by one polynomial evaluation and one scalar multiplication instead of an MSM.
 * It exists for the 2^22-gate parity test (BASELINE config 5), where eleven CPU MSMs of 4 M terms would take minutes;
 * tests/test_oracle_c_prove.py checks that both commit paths give the same proof bytes.  Call before the first prove;
 * the VerifierKey commitments of oracle_prover_new are unaffected (pass vk48, or let them be MSMs). */
void oracle_prover_set_trapdoor(oprover* P, const u64 tau_m[4], const u64 g_scalar_m[4]) {
  P->trapdoor = 1; P->tau = fr_ld(tau_m); P->gscalar = fr_ld(g_scalar_m);
}

void oracle_prover_set_version(void* h, int version) { ((oprover*)h)->version = version; }
/* The 15 VerifierKey commitments (Compiler::preprocess, compiler.rs:213-232) recomputed through the trapdoor — for sizes where
 * the caller handed oracle_prover_new a vk48 it did not compute itself (the 2^22-gate test takes the GPU's) and 15 CPU MSMs
 * of 4 M terms are out of reach: the test compares these bytes with the ones it passed in.  -1 without a trapdoor. */
int oracle_prover_vk_trapdoor(const oprover* P, uint8_t out[15 * 48]);
/* ... and adopted as the prover's own VerifierKey (what seeds its transcript): an oracle_prover_new that was handed a
 * placeholder vk48 becomes independent of whoever produced one — the 2^22-gate test runs this oracle BESIDE the GPU. */
int oracle_prover_adopt_vk_trapdoor(oprover* P) { return oracle_prover_vk_trapdoor(P, (uint8_t*)P->vk); }
int oracle_prover_vk_trapdoor(const oprover* P, uint8_t out[15 * 48]) {
  if (!P->trapdoor) return -1;
  for (int k = 0; k < K_COUNT; ++k) { const int rc = commit48(P, P->polys[k], P->poly_len[k], out + 48 * k); if (rc) return rc; }
  return 0;
}
static void commit_trapdoor(const oprover* P, const fr* poly, u64 len, uint8_t raw[97]) {
  memset(raw, 0, 97);
  const u64 chunk = 1 << 14, nch = (len + chunk - 1) / chunk;
  fr total = fr_zero();
  if (nch) {
    fr* part = (fr*)malloc(32 * nch);
#pragma omp parallel for num_threads(P->threads) schedule(static)
    for (u64 c = 0; c < nch; ++c) {                                  /* Horner inside a chunk */
      const u64 lo = c * chunk, hi = lo + chunk < len ? lo + chunk : len;
      fr acc = fr_zero();
      for (u64 i = hi; i-- > lo;) acc = fr_add(fr_mul(acc, P->tau), poly[i]);
      part[c] = acc;
    }
    const fr step = fr_pow(P->tau, chunk);
    for (u64 c = nch; c-- > 0;) total = fr_add(fr_mul(total, step), part[c]);
    free(part);
  }
  uint8_t kb[32];
  fr_to_bytes(fr_mul(total, P->gscalar), kb);                        /* g p(tau), canonical little-endian */
  g1a gen; memcpy(gen.x.l, G1_GEN_X, 48); memcpy(gen.y.l, G1_GEN_Y, 48);
  g1j acc = j_identity();
  for (int bit = 255; bit >= 0; --bit) {
    acc = j_double(acc);
    if ((kb[bit >> 3] >> (bit & 7)) & 1) acc = j_add_mixed(acc, &gen);
  }
  if (fp_is_zero(acc.z)) { raw[96] = 1; return; }
  g1a a;
  j_to_affine(acc, &a);
  memcpy(raw, a.x.l, 48); memcpy(raw + 48, a.y.l, 48);
}

/* ------------------------------------------------------------------ widget identities */
typedef struct { fr a, b, c, d, a_w, b_w, d_w, q_l, q_r, q_c; } wvals;
static fr C2, C3, C4, C9, C18, C81, C83, EDW_D;   /* set once in consts_init */
static int consts_ready = 0;
static void consts_init(void) {
  if (consts_ready) return;
  C2 = fr_small(2); C3 = fr_small(3); C4 = fr_small(4); C9 = fr_small(9); C18 = fr_small(18); C81 = fr_small(81); C83 = fr_small(83);
  EDW_D = fr_neg(fr_mul(fr_small(10240), fr_inv(fr_small(10241))));   /* dusk_jubjub::EDWARDS_D */
  consts_ready = 1;
}
static inline fr mul4(fr a) { return fr_dbl(fr_dbl(a)); }
static fr delta(fr f) {   /* range/proverkey.rs:88-93: f (f-1)(f-2)(f-3) */
  const fr one = fr_one();
  fr f1 = fr_sub(f, one), f2 = fr_sub(f1, one), f3 = fr_sub(f2, one);
  return fr_mul(fr_mul(f, f1), fr_mul(f2, f3));
}
static fr range_identity(fr ch, const wvals* v) {   /* range/proverkey.rs:32-58 */
  const fr k = fr_sqr(ch), k2 = fr_sqr(k), k3 = fr_mul(k2, k);
  fr r = delta(fr_sub(v->c, mul4(v->d)));
  r = fr_add(r, fr_mul(delta(fr_sub(v->b, mul4(v->c))), k));
  r = fr_add(r, fr_mul(delta(fr_sub(v->a, mul4(v->b))), k2));
  r = fr_add(r, fr_mul(delta(fr_sub(v->d_w, mul4(v->a))), k3));
  return r;
}
static fr delta_xor_and(fr a, fr b, fr w, fr c, fr q_c) {   /* logic/proverkey.rs:108-144 */
  const fr ab = fr_add(a, b);
  /* F = w (w (4w - 18(a+b) + 81) + 18(a^2 + b^2) - 81(a+b) + 83) */
  fr inner = fr_add(fr_sub(mul4(w), fr_mul(C18, ab)), C81);
  fr F = fr_mul(w, inner);
  F = fr_add(F, fr_mul(C18, fr_add(fr_sqr(a), fr_sqr(b))));
  F = fr_sub(F, fr_mul(C81, ab));
  F = fr_add(F, C83);
  F = fr_mul(w, F);
  const fr E = fr_sub(fr_mul(C3, fr_add(ab, c)), fr_dbl(F));
  const fr B = fr_mul(q_c, fr_sub(fr_mul(C9, c), fr_mul(C3, ab)));
  return fr_add(B, E);
}
static fr logic_identity(fr ch, const wvals* v) {   /* logic/proverkey.rs:34-70 */
  const fr k = fr_sqr(ch), k2 = fr_sqr(k), k3 = fr_mul(k2, k), k4 = fr_mul(k3, k);
  const fr a = fr_sub(v->a_w, mul4(v->a)), b = fr_sub(v->b_w, mul4(v->b)), d = fr_sub(v->d_w, mul4(v->d)), w = v->c;
  fr r = delta(a);
  r = fr_add(r, fr_mul(delta(b), k));
  r = fr_add(r, fr_mul(delta(d), k2));
  r = fr_add(r, fr_mul(fr_sub(w, fr_mul(a, b)), k3));
  r = fr_add(r, fr_mul(delta_xor_and(a, b, w, d, v->q_c), k4));
  return r;
}
static fr fixed_identity(fr ch, const wvals* v) {   /* ecc/scalar_mul/fixed_base/proverkey.rs:39-101 */
  const fr one = fr_one();
  const fr k = fr_sqr(ch), k2 = fr_sqr(k), k3 = fr_mul(k2, k);
  const fr x_beta = v->q_l, y_beta = v->q_r;
  const fr bit = fr_sub(fr_sub(v->d_w, v->d), v->d);
  const fr bit_consistency = fr_mul(fr_mul(bit, fr_sub(bit, one)), fr_add(bit, one));
  const fr y_alpha = fr_add(fr_mul(fr_sqr(bit), fr_sub(y_beta, one)), one);
  const fr x_alpha = fr_mul(bit, x_beta);
  const fr xy_consistency = fr_mul(fr_sub(fr_mul(bit, v->q_c), v->c), k);
  const fr cabd = fr_mul(fr_mul(fr_mul(v->c, v->a), v->b), EDW_D);
  const fr x_acc = fr_mul(fr_sub(fr_add(v->a_w, fr_mul(v->a_w, cabd)), fr_add(fr_mul(v->a, y_alpha), fr_mul(v->b, x_alpha))), k2);
  const fr y_acc = fr_mul(fr_sub(fr_sub(v->b_w, fr_mul(v->b_w, cabd)), fr_add(fr_mul(v->b, y_alpha), fr_mul(v->a, x_alpha))), k3);
  return fr_add(fr_add(bit_consistency, x_acc), fr_add(y_acc, xy_consistency));
}
static fr var_identity(fr ch, const wvals* v) {     /* ecc/curve_addition/proverkey.rs:33-77 */
  const fr k = fr_sqr(ch);
  const fr x_1 = v->a, x_3 = v->a_w, y_1 = v->b, y_3 = v->b_w, x_2 = v->c, y_2 = v->d, x1_y2 = v->d_w;
  const fr xy_consistency = fr_sub(fr_mul(x_1, y_2), x1_y2);
  const fr y1_x2 = fr_mul(y_1, x_2), y1_y2 = fr_mul(y_1, y_2), x1_x2 = fr_mul(x_1, x_2);
  const fr dxy = fr_mul(fr_mul(EDW_D, x1_y2), y1_x2);
  const fr x3c = fr_mul(fr_sub(fr_add(x1_y2, y1_x2), fr_add(x_3, fr_mul(x_3, dxy))), k);
  const fr y3c = fr_mul(fr_sub(fr_add(y1_y2, x1_x2), fr_sub(y_3, fr_mul(y_3, dxy))), fr_sqr(k));
  return fr_add(fr_add(xy_consistency, x3c), y3c);
}

/* ------------------------------------------------------------------ polynomial helpers */
/* Polynomial::evaluate (polynomial.rs:120-137): chunked Horner */
static fr poly_eval(const fr* p, u64 len, fr x, int threads) {
  if (len == 0) return fr_zero();
  const u64 chunk = 1 << 14;
  const u64 nch = (len + chunk - 1) / chunk;
  fr* part = (fr*)malloc(32 * nch);
#pragma omp parallel for num_threads(threads) schedule(static) if (nch > 1)
  for (u64 c = 0; c < nch; ++c) {
    const u64 lo = c * chunk, hi = lo + chunk < len ? lo + chunk : len;
    fr acc = fr_zero();
    for (u64 i = hi; i-- > lo;) acc = fr_add(fr_mul(acc, x), p[i]);
    part[c] = acc;
  }
  const fr xc = fr_pow(x, chunk);
  fr acc = fr_zero();
  for (u64 c = nch; c-- > 0;) acc = fr_add(fr_mul(acc, xc), part[c]);
  free(part);
  return acc;
}
/* ruffini (polynomial.rs:345-367): quotient of p(X) / (X - z), remainder dropped; out has len - 1 entries */
static void poly_ruffini(const fr* p, u64 len, fr z, fr* out) {
  fr k = fr_zero();
  for (u64 i = len; i-- > 0;) {
    const fr t = fr_add(p[i], k);
    if (i > 0) out[i - 1] = t;
    k = fr_mul(z, t);
  }
}
/* util::batch_inversion (util.rs:87-117), zeros skipped */
static void batch_inverse(fr* v, u64 len, int threads) {
  const u64 chunk = 1 << 12;
  const u64 nch = (len + chunk - 1) / chunk;
#pragma omp parallel for num_threads(threads) schedule(static) if (nch > 1)
  for (u64 c = 0; c < nch; ++c) {
    const u64 lo = c * chunk, hi = lo + chunk < len ? lo + chunk : len;
    fr pre[1 << 12];
    fr acc = fr_one();
    for (u64 i = lo; i < hi; ++i) { pre[i - lo] = acc; if (!fr_is_zero(v[i])) acc = fr_mul(acc, v[i]); }
    fr inv = fr_inv(acc);
    for (u64 i = hi; i-- > lo;) {
      if (fr_is_zero(v[i])) continue;
      const fr t = fr_mul(inv, pre[i - lo]);
      inv = fr_mul(inv, v[i]);
      v[i] = t;
    }
  }
}

/* ------------------------------------------------------------------ prove */
typedef struct {   /* optional stage outputs; any pointer may be NULL */
  u64* wire_polys;   /* 4 x (n + 8): blinded a, b, c, d coefficients */
  u64* z_poly;       /* n + 8 */
  u64* t_poly;       /* 8n: quotient coefficients (before the split / blinding) */
  u64* w_z;          /* n + 8 */
  u64* w_zw;         /* n + 8 */
  u64* evals;        /* 15 x 4: Proof order */
  u64* challenges;   /* 10 x 4: beta gamma alpha range logic fixed var z v v_w */
  double* seconds;   /* 6: ntt, msm, quotient point-wise, grand product, round 4-5 O(n), total */
} oracle_trace;

static double now_s(void) { return omp_get_wtime(); }

/* Returns 0, -3 (PolynomialDegreeTooLarge), -6 (CircuitUnsatisfied), -1 (argument). */
int oracle_prover_prove(const oprover* P, const u64* const wires[4], const u64* pi_idx, const u64* pi_val, u64 pi_count,
                        const u64* blinders, uint8_t proof[1008], oracle_trace* trc) {
  consts_init();
  const u64 n = P->n, n8 = P->n8, np = n + 8;
  const uint32_t L = P->logn;
  const int T = P->threads;
  const fr one = fr_one();
  const fr omega = omega_of(L);
  double t_ntt = 0, t_msm = 0, t_quot = 0, t_perm = 0, t_tail = 0, t0, t_start = now_s();
  int rc = 0;
#define BL(i) fr_ld(blinders + 4 * (i))

  strobe tr;
  tr_init(&tr, P->label, P->label_len);
  tr_domain_sep(&tr, P->constraints);
  /* V2 (Transcript::base + seed_transcript_legacy, transcript.rs:110-129, widget.rs:224-228,260-265): the label s_sigma_4 carries s_sigma_1's commitment */
  for (int k = 0; k < 15; ++k) tr_append(&tr, VK_LABEL[k], P->vk[(P->version == 2 && k == 14) ? K_S1 : VK_ORDER[k]], 48);
  tr_domain_sep(&tr, P->constraints);
  for (u64 i = 0; i < pi_count; ++i) tr_scalar(&tr, "pi", fr_ld(pi_val + 4 * i));

  uint8_t comm[11][48];
  /* ---- round 1 (prover.rs:444-479): blinded wire polynomials */
  fr* wp[4];
  t0 = now_s();
  for (int k = 0; k < 4; ++k) {
    wp[k] = (fr*)calloc(np, 32);
    memcpy(wp[k], wires[k], 32 * n);
    oracle_ntt((u64*)wp[k], L, 1, 0, n, T);
    for (int j = 0; j < 2; ++j) { const fr b = BL(2 * k + j); wp[k][j] = fr_sub(wp[k][j], b); wp[k][n + j] = b; }   /* blind_poly :139-152 */
  }
  t_ntt += now_s() - t0;
  t0 = now_s();
  for (int k = 0; k < 4 && !rc; ++k) rc = commit48(P, wp[k], n + 2, comm[k]);
  t_msm += now_s() - t0;
  u64 wz_len = 0;
  fr *zp = NULL, *pip = NULL, *cos[6] = {0}, *tq = NULL, *agg = NULL, *wz = NULL, *wzw = NULL, *num = NULL, *den = NULL;
  if (rc) goto done;
  tr_append(&tr, "a_comm", comm[0], 48); tr_append(&tr, "b_comm", comm[1], 48);
  tr_append(&tr, "c_comm", comm[2], 48); tr_append(&tr, "d_comm", comm[3], 48);

  /* ---- round 2 (prover.rs:481-505): permutation polynomial (permutation.rs:213-294) */
  const fr beta = tr_challenge(&tr, "beta");
  tr_scalar(&tr, "beta", beta);
  const fr gamma = tr_challenge(&tr, "gamma");
  t0 = now_s();
  num = (fr*)malloc(32 * n); den = (fr*)malloc(32 * n);
  {
    const fr ks[4] = {one, fr_small(7), fr_small(13), fr_small(17)};
    const u64 chunk = 1 << 12;
#pragma omp parallel for num_threads(T) schedule(static)
    for (u64 c0 = 0; c0 < n; c0 += chunk) {
      fr root = fr_pow(omega, c0);
      for (u64 i = c0; i < c0 + chunk && i < n; ++i) {
        fr nu = one, de = one;
        const fr br = fr_mul(beta, root);
        for (int k = 0; k < 4; ++k) {
          const fr w = fr_ld(wires[k] + 4 * i);
          nu = fr_mul(nu, fr_add(fr_add(w, fr_mul(br, ks[k])), gamma));
          de = fr_mul(de, fr_add(fr_add(w, fr_mul(beta, P->sigma_n[k][i])), gamma));
        }
        num[i] = nu; den[i] = de;
        root = fr_mul(root, omega);
      }
    }
    for (u64 i = 0; i + 1 < n; ++i) if (fr_is_zero(den[i])) { rc = -1; }   /* "permutation denominator must be nonzero" */
    if (rc) goto done;
    batch_inverse(den, n, T);
    zp = (fr*)calloc(np, 32);
    fr prod = one;
    for (u64 i = 0; i < n; ++i) { zp[i] = prod; if (i + 1 < n) prod = fr_mul(prod, fr_mul(num[i], den[i])); }
  }
  t_perm += now_s() - t0;
  t0 = now_s();
  oracle_ntt((u64*)zp, L, 1, 0, n, T);
  t_ntt += now_s() - t0;
  for (int j = 0; j < 3; ++j) { const fr b = BL(8 + j); zp[j] = fr_sub(zp[j], b); zp[n + j] = b; }
  t0 = now_s();
  rc = commit48(P, zp, n + 3, comm[4]);
  t_msm += now_s() - t0;
  if (rc) goto done;
  tr_append(&tr, "z_comm", comm[4], 48);

  /* ---- round 3 (prover.rs:507-589): quotient (quotient_poly.rs:20-137) */
  const fr alpha = tr_challenge(&tr, "alpha");
  const fr range_ch = tr_challenge(&tr, "range separation challenge");
  const fr logic_ch = tr_challenge(&tr, "logic separation challenge");
  const fr fixed_ch = tr_challenge(&tr, "fixed base separation challenge");
  const fr var_ch = tr_challenge(&tr, "variable base separation challenge");
  t0 = now_s();
  pip = (fr*)calloc(np, 32);
  for (u64 i = 0; i < pi_count; ++i) { if (pi_idx[i] >= n) { rc = -1; goto done; } pip[pi_idx[i]] = fr_ld(pi_val + 4 * i); }
  if (pi_count) oracle_ntt((u64*)pip, L, 1, 0, n, T);
  {
    const fr* src[6] = {zp, wp[0], wp[1], wp[2], wp[3], pip};
    const u64 len[6] = {n + 3, n + 2, n + 2, n + 2, n + 2, n};
    for (int k = 0; k < 6; ++k) {                       /* quotient_poly.rs:139-157,177 */
      cos[k] = (fr*)calloc(n8, 32);
      memcpy(cos[k], src[k], 32 * len[k]);
      oracle_ntt((u64*)cos[k], L + 3, 0, 1, len[k], T);
    }
  }
  t_ntt += now_s() - t0;
  t0 = now_s();
  tq = (fr*)malloc(32 * n8);
  {
    const fr* z8 = cos[0]; const fr* a8 = cos[1]; const fr* b8 = cos[2]; const fr* c8 = cos[3]; const fr* d8 = cos[4]; const fr* pi8 = cos[5];
    fr* lag = (fr*)malloc(32 * n8);                      /* L1 over the coset (quotient_poly.rs:266-284) */
#pragma omp parallel for num_threads(T) schedule(static)
    for (u64 i = 0; i < n8; ++i) lag[i] = fr_sub(P->ev8[K_COUNT][i], one);
    batch_inverse(lag, n8, T);
    const fr n_inv = fr_inv(fr_from_u64(n));
    const fr l1_alpha_sq = fr_sqr(alpha);
    const fr ks[4] = {one, fr_small(7), fr_small(13), fr_small(17)};
    const int has_range = P->poly_len[K_QRANGE] != 0, has_logic = P->poly_len[K_QLOGIC] != 0, has_fixed = P->poly_len[K_QFIXED] != 0,
              has_var = P->poly_len[K_QVAR] != 0;
#pragma omp parallel for num_threads(T) schedule(static)
    for (u64 i = 0; i < n8; ++i) {
      const u64 iw = (i + 8) & (n8 - 1);                 /* rotation by one row of the proving domain (:61-67,191-193) */
      wvals v;
      v.a = a8[i]; v.b = b8[i]; v.c = c8[i]; v.d = d8[i]; v.a_w = a8[iw]; v.b_w = b8[iw]; v.d_w = d8[iw];
      v.q_l = P->ev8[K_QL][i]; v.q_r = P->ev8[K_QR][i]; v.q_c = P->ev8[K_QC][i];
      /* arithmetic/proverkey.rs:44-71 */
      fr t1 = fr_mul(fr_mul(v.a, v.b), P->ev8[K_QM][i]);
      t1 = fr_add(t1, fr_mul(v.a, v.q_l));
      t1 = fr_add(t1, fr_mul(v.b, v.q_r));
      t1 = fr_add(t1, fr_mul(v.c, P->ev8[K_QO][i]));
      t1 = fr_add(t1, fr_mul(v.d, P->ev8[K_QF][i]));
      t1 = fr_add(t1, v.q_c);
      t1 = fr_mul(t1, P->ev8[K_QARITH][i]);
      /* a selector polynomial that is identically zero contributes 0 * identity: skipped */
      if (has_range) t1 = fr_add(t1, fr_mul(fr_mul(range_identity(range_ch, &v), P->ev8[K_QRANGE][i]), range_ch));
      if (has_logic) t1 = fr_add(t1, fr_mul(fr_mul(P->ev8[K_QLOGIC][i], logic_identity(logic_ch, &v)), logic_ch));
      if (has_fixed) t1 = fr_add(t1, fr_mul(fr_mul(fixed_identity(fixed_ch, &v), P->ev8[K_QFIXED][i]), fixed_ch));
      if (has_var) t1 = fr_add(t1, fr_mul(fr_mul(var_identity(var_ch, &v), P->ev8[K_QVAR][i]), var_ch));
      t1 = fr_add(t1, pi8[i]);
      /* permutation/proverkey.rs:40-125 */
      const fr x = P->ev8[K_COUNT][i];
      const fr bx = fr_mul(beta, x);
      fr ident = fr_add(fr_add(v.a, bx), gamma);
      ident = fr_mul(ident, fr_add(fr_add(v.b, fr_mul(bx, ks[1])), gamma));
      ident = fr_mul(ident, fr_add(fr_add(v.c, fr_mul(bx, ks[2])), gamma));
      ident = fr_mul(ident, fr_add(fr_add(v.d, fr_mul(bx, ks[3])), gamma));
      ident = fr_mul(fr_mul(ident, z8[i]), alpha);
      fr copy = fr_add(fr_add(v.a, fr_mul(beta, P->ev8[K_S1][i])), gamma);
      copy = fr_mul(copy, fr_add(fr_add(v.b, fr_mul(beta, P->ev8[K_S2][i])), gamma));
      copy = fr_mul(copy, fr_add(fr_add(v.c, fr_mul(beta, P->ev8[K_S3][i])), gamma));
      copy = fr_mul(copy, fr_add(fr_add(v.d, fr_mul(beta, P->ev8[K_S4][i])), gamma));
      copy = fr_mul(fr_mul(copy, z8[iw]), alpha);
      const fr l1 = fr_mul(fr_mul(lag[i], P->vh[i & 7]), n_inv);
      const fr onec = fr_mul(fr_sub(z8[i], one), fr_mul(l1, l1_alpha_sq));
      const fr t2 = fr_add(fr_sub(ident, copy), onec);
      tq[i] = fr_mul(fr_add(t1, t2), P->vinv[i & 7]);     /* :96-101 */
    }
    free(lag);
  }
  t_quot += now_s() - t0;
  t0 = now_s();
  oracle_ntt((u64*)tq, L + 3, 1, 1, n8, T);               /* coset_ifft :103 */
  t_ntt += now_s() - t0;
  const u64 tlen = trimmed_len(tq, n8);
  if (tlen > 7 * n) { rc = -6; goto done; }               /* Error::CircuitUnsatisfied, quotient_poly.rs:132 */
  if (trc && trc->t_poly) memcpy(trc->t_poly, tq, 32 * n8);
  /* split + blinding (prover.rs:547-574) */
  fr* tpart[4];
  u64 tplen[4];
  for (int k = 0; k < 4; ++k) tpart[k] = (fr*)calloc(k == 3 ? 5 * n + 8 : np, 32);
  memcpy(tpart[0], tq, 32 * n); memcpy(tpart[1], tq + n, 32 * n); memcpy(tpart[2], tq + 2 * n, 32 * n);
  {
    u64 l4 = tlen > 3 * n ? tlen - 3 * n : 0;
    memcpy(tpart[3], tq + 3 * n, 32 * l4);
    const fr b12 = BL(11), b13 = BL(12), b14 = BL(13);
    tpart[0][n] = b12;
    tpart[1][0] = fr_sub(tpart[1][0], b12); tpart[1][n] = b13;
    tpart[2][0] = fr_sub(tpart[2][0], b13); tpart[2][n] = b14;
    tpart[3][0] = fr_sub(tpart[3][0], b14);
    tplen[0] = tplen[1] = tplen[2] = n + 1;
    tplen[3] = l4 ? l4 : 1;
  }
  t0 = now_s();
  for (int k = 0; k < 4 && !rc; ++k) rc = commit48(P, tpart[k], tplen[k], comm[5 + k]);
  t_msm += now_s() - t0;
  if (rc) { for (int k = 0; k < 4; ++k) free(tpart[k]); goto done; }
  tr_append(&tr, "t_low_comm", comm[5], 48); tr_append(&tr, "t_mid_comm", comm[6], 48);
  tr_append(&tr, "t_high_comm", comm[7], 48); tr_append(&tr, "t_fourth_comm", comm[8], 48);

  /* ---- round 4 (prover.rs:591-676) */
  const fr z_ch = tr_challenge(&tr, "z_challenge");
  const fr zw = fr_mul(z_ch, omega);
  t0 = now_s();
  const fr e_a = poly_eval(wp[0], n + 2, z_ch, T), e_b = poly_eval(wp[1], n + 2, z_ch, T), e_c = poly_eval(wp[2], n + 2, z_ch, T),
           e_d = poly_eval(wp[3], n + 2, z_ch, T);
  const fr e_aw = poly_eval(wp[0], n + 2, zw, T), e_bw = poly_eval(wp[1], n + 2, zw, T), e_dw = poly_eval(wp[3], n + 2, zw, T);
  const fr e_qarith = poly_eval(P->polys[K_QARITH], P->poly_len[K_QARITH], z_ch, T), e_qc = poly_eval(P->polys[K_QC], P->poly_len[K_QC], z_ch, T),
           e_ql = poly_eval(P->polys[K_QL], P->poly_len[K_QL], z_ch, T), e_qr = poly_eval(P->polys[K_QR], P->poly_len[K_QR], z_ch, T);
  const fr e_s1 = poly_eval(P->polys[K_S1], P->poly_len[K_S1], z_ch, T), e_s2 = poly_eval(P->polys[K_S2], P->poly_len[K_S2], z_ch, T),
           e_s3 = poly_eval(P->polys[K_S3], P->poly_len[K_S3], z_ch, T);
  const fr e_z = poly_eval(zp, n + 3, zw, T);
  t_tail += now_s() - t0;
  tr_scalar(&tr, "a_eval", e_a); tr_scalar(&tr, "b_eval", e_b); tr_scalar(&tr, "c_eval", e_c); tr_scalar(&tr, "d_eval", e_d);
  tr_scalar(&tr, "s_sigma_1_eval", e_s1); tr_scalar(&tr, "s_sigma_2_eval", e_s2); tr_scalar(&tr, "s_sigma_3_eval", e_s3);
  tr_scalar(&tr, "z_eval", e_z);
  tr_scalar(&tr, "a_w_eval", e_aw); tr_scalar(&tr, "b_w_eval", e_bw); tr_scalar(&tr, "d_w_eval", e_dw);
  tr_scalar(&tr, "q_arith_eval", e_qarith); tr_scalar(&tr, "q_c_eval", e_qc); tr_scalar(&tr, "q_l_eval", e_ql); tr_scalar(&tr, "q_r_eval", e_qr);

  /* ---- round 5 (prover.rs:678-739) */
  const fr v_ch = tr_challenge(&tr, "v_challenge");
  t0 = now_s();
  {
    wvals ev;
    ev.a = e_a; ev.b = e_b; ev.c = e_c; ev.d = e_d; ev.a_w = e_aw; ev.b_w = e_bw; ev.d_w = e_dw; ev.q_l = e_ql; ev.q_r = e_qr; ev.q_c = e_qc;
    const fr z_n = fr_pow(z_ch, n);
    const fr zh = fr_sub(z_n, one);
    const fr n_inv = fr_inv(fr_from_u64(n));
    /* compute_barycentric_eval (proof.rs:1041-1088) */
    fr pi_eval = fr_zero();
    if (pi_count) {
      const fr omega_inv = fr_inv(omega);
      fr acc = fr_zero();
      for (u64 i = 0; i < pi_count; ++i) {
        const fr val = fr_ld(pi_val + 4 * i);
        if (fr_is_zero(val)) continue;
        const fr dn = fr_sub(fr_mul(fr_pow(omega_inv, pi_idx[i]), z_ch), one);
        acc = fr_add(acc, fr_mul(fr_inv(dn), val));
      }
      pi_eval = fr_mul(acc, fr_mul(zh, n_inv));
    }
    /* permutation/proverkey.rs:127-269 */
    const fr bz = fr_mul(beta, z_ch);
    fr lin_a = fr_add(fr_add(e_a, bz), gamma);
    lin_a = fr_mul(lin_a, fr_add(fr_add(e_b, fr_mul(fr_small(7), bz)), gamma));
    lin_a = fr_mul(lin_a, fr_add(fr_add(e_c, fr_mul(fr_small(13), bz)), gamma));
    lin_a = fr_mul(lin_a, fr_add(fr_add(e_d, fr_mul(fr_small(17), bz)), gamma));
    lin_a = fr_mul(lin_a, alpha);
    fr lin_b = fr_add(fr_add(e_a, fr_mul(beta, e_s1)), gamma);
    lin_b = fr_mul(lin_b, fr_add(fr_add(e_b, fr_mul(beta, e_s2)), gamma));
    lin_b = fr_mul(lin_b, fr_add(fr_add(e_c, fr_mul(beta, e_s3)), gamma));
    lin_b = fr_mul(fr_mul(lin_b, fr_mul(beta, e_z)), alpha);
    fr l1_z;                                               /* evaluate_all_lagrange_coefficients(z)[0], domain.rs:237-284 */
    if (fr_eq(z_n, one)) l1_z = fr_eq(z_ch, one) ? one : fr_zero();
    else l1_z = fr_mul(fr_mul(zh, n_inv), fr_inv(fr_sub(z_ch, one)));
    const fr c_range = fr_mul(range_identity(range_ch, &ev), range_ch);
    const fr c_logic = fr_mul(logic_identity(logic_ch, &ev), logic_ch);
    const fr c_fixed = fr_mul(fixed_identity(fixed_ch, &ev), fixed_ch);
    const fr c_var = fr_mul(var_identity(var_ch, &ev), var_ch);
    const fr nzh = fr_neg(zh);
    fr vp[12];
    vp[0] = one;
    for (int k = 1; k < 12; ++k) vp[k] = fr_mul(vp[k - 1], v_ch);
    /* r(X) (linearization_poly.rs:168-264) folded with the aggregate-witness sum (key.rs:394-417) over
     * [r, a, b, c, d, s1, s2, s3, q_arith, q_c, q_l, q_r] (prover.rs:706-726) */
    struct { const fr* p; u64 len; fr s; } term[24];
    int nt = 0;
#define TERM(ptr, l, sc) do { term[nt].p = (ptr); term[nt].len = (l); term[nt].s = (sc); ++nt; } while (0)
    TERM(P->polys[K_QM], P->poly_len[K_QM], fr_mul(e_qarith, fr_mul(e_a, e_b)));
    TERM(P->polys[K_QL], P->poly_len[K_QL], fr_add(fr_mul(e_qarith, e_a), vp[10]));
    TERM(P->polys[K_QR], P->poly_len[K_QR], fr_add(fr_mul(e_qarith, e_b), vp[11]));
    TERM(P->polys[K_QO], P->poly_len[K_QO], fr_mul(e_qarith, e_c));
    TERM(P->polys[K_QF], P->poly_len[K_QF], fr_mul(e_qarith, e_d));
    TERM(P->polys[K_QC], P->poly_len[K_QC], fr_add(e_qarith, vp[9]));
    TERM(P->polys[K_QARITH], P->poly_len[K_QARITH], vp[8]);
    TERM(P->polys[K_QRANGE], P->poly_len[K_QRANGE], c_range);
    TERM(P->polys[K_QLOGIC], P->poly_len[K_QLOGIC], c_logic);
    TERM(P->polys[K_QFIXED], P->poly_len[K_QFIXED], c_fixed);
    TERM(P->polys[K_QVAR], P->poly_len[K_QVAR], c_var);
    TERM(P->polys[K_S1], P->poly_len[K_S1], vp[5]);
    TERM(P->polys[K_S2], P->poly_len[K_S2], vp[6]);
    TERM(P->polys[K_S3], P->poly_len[K_S3], vp[7]);
    TERM(P->polys[K_S4], P->poly_len[K_S4], fr_neg(lin_b));
    TERM(zp, n + 3, fr_add(lin_a, fr_mul(l1_z, fr_sqr(alpha))));
    TERM(wp[0], n + 2, vp[1]); TERM(wp[1], n + 2, vp[2]); TERM(wp[2], n + 2, vp[3]); TERM(wp[3], n + 2, vp[4]);
    TERM(tpart[0], tplen[0], nzh);
    TERM(tpart[1], tplen[1], fr_mul(nzh, z_n));
    TERM(tpart[2], tplen[2], fr_mul(nzh, fr_sqr(z_n)));
    TERM(tpart[3], tplen[3], fr_mul(nzh, fr_mul(fr_sqr(z_n), z_n)));
#undef TERM
    const u64 alen = tplen[3] > np - 1 ? tplen[3] : np - 1;   /* honest proofs: every term has at most n + 7 coefficients */
    agg = (fr*)calloc(alen + 8, 32);
#pragma omp parallel for num_threads(T) schedule(static)
    for (u64 i = 0; i < alen; ++i) {
      fr acc = i == 0 ? pi_eval : fr_zero();
      for (int k = 0; k < nt; ++k) if (i < term[k].len) acc = fr_add(acc, fr_mul(term[k].p[i], term[k].s));
      agg[i] = acc;
    }
    wz = (fr*)calloc(alen + 8, 32);
    wz_len = alen - 1;
    poly_ruffini(agg, alen, z_ch, wz);
  }
  t_tail += now_s() - t0;
  t0 = now_s();
  rc = commit48(P, wz, wz_len, comm[9]);
  t_msm += now_s() - t0;
  for (int k = 0; k < 4; ++k) free(tpart[k]);
  if (rc) goto done;
  const fr v_w = tr_challenge(&tr, "v_w_challenge");
  t0 = now_s();
  {
    const fr vw2 = fr_sqr(v_w), vw3 = fr_mul(vw2, v_w);
    const u64 alen = np - 1;
#pragma omp parallel for num_threads(T) schedule(static)
    for (u64 i = 0; i < alen; ++i) {
      fr acc = i < n + 3 ? zp[i] : fr_zero();
      if (i < n + 2) {
        acc = fr_add(acc, fr_mul(wp[0][i], v_w));
        acc = fr_add(acc, fr_mul(wp[1][i], vw2));
        acc = fr_add(acc, fr_mul(wp[3][i], vw3));
      }
      agg[i] = acc;
    }
    wzw = (fr*)calloc(np, 32);
    poly_ruffini(agg, alen, zw, wzw);
  }
  t_tail += now_s() - t0;
  t0 = now_s();
  rc = commit48(P, wzw, np - 2, comm[10]);
  t_msm += now_s() - t0;
  if (rc) goto done;

  /* ---- Proof::to_bytes (proof.rs:137-162, linearization_poly.rs:98-124) */
  memcpy(proof, comm, 11 * 48);
  {
    const fr order[15] = {e_a, e_b, e_c, e_d, e_aw, e_bw, e_dw, e_qarith, e_qc, e_ql, e_qr, e_s1, e_s2, e_s3, e_z};
    for (int k = 0; k < 15; ++k) fr_to_bytes(order[k], proof + 11 * 48 + 32 * k);
    if (trc && trc->evals) memcpy(trc->evals, order, sizeof order);
  }
  if (trc) {
    if (trc->wire_polys) for (int k = 0; k < 4; ++k) memcpy(trc->wire_polys + 4 * np * k, wp[k], 32 * np);
    if (trc->z_poly) memcpy(trc->z_poly, zp, 32 * np);
    if (trc->w_z) memcpy(trc->w_z, wz, 32 * np);
    if (trc->w_zw) memcpy(trc->w_zw, wzw, 32 * np);
    if (trc->challenges) {
      const fr ch[10] = {beta, gamma, alpha, range_ch, logic_ch, fixed_ch, var_ch, z_ch, v_ch, v_w};
      memcpy(trc->challenges, ch, sizeof ch);
    }
  }
done:
  if (trc && trc->seconds) {
    trc->seconds[0] = t_ntt; trc->seconds[1] = t_msm; trc->seconds[2] = t_quot; trc->seconds[3] = t_perm; trc->seconds[4] = t_tail;
    trc->seconds[5] = now_s() - t_start;
  }
  for (int k = 0; k < 4; ++k) free(wp[k]);
  for (int k = 0; k < 6; ++k) free(cos[k]);
  free(zp); free(pip); free(tq); free(agg); free(wz); free(wzw); free(num); free(den);
  return rc;
#undef BL
}

/* ------------------------------------------------------------------ synthetic SRS */
/* PublicParameters::setup semantics (srs.rs:61-100): out[i] = (g_scalar * tau^i) * G1::generator as 96-byte
 * x || y Montgomery points.  Fixed-base windowed multiplication (8-bit windows, 32 x 255 table)
 * instead of the reference's per-point double-and-add (util.rs:77) — same points. */

int oracle_srs_generate(const u64 tau_m[4], const u64 g_scalar_m[4], u64 n, uint8_t* out96, int threads) {
  if (threads <= 0) threads = omp_get_max_threads();
  g1a* table = (g1a*)malloc(sizeof(g1a) * 32 * 255);   /* table[w][d-1] = d * 2^(8w) * G */
  g1a base; memcpy(base.x.l, G1_GEN_X, 48); memcpy(base.y.l, G1_GEN_Y, 48);
  for (int w = 0; w < 32; ++w) {          /* d * base for d = 1 .. 256, normalised with one shared inversion per window */
    g1j mult[256];
    fp pre[256], run;
    memcpy(run.l, FP_ONE, 48);
    g1j acc = j_identity();
    for (int d = 0; d < 256; ++d) { acc = j_add_mixed(acc, &base); mult[d] = acc; pre[d] = run; run = fp_mul(run, acc.z); }
    fp inv = fp_inv(run);
    g1a norm[256];
    for (int d = 256; d-- > 0;) {
      const fp zi = fp_mul(inv, pre[d]), zi2 = fp_mul(zi, zi);
      inv = fp_mul(inv, mult[d].z);
      norm[d].x = fp_mul(mult[d].x, zi2);
      norm[d].y = fp_mul(mult[d].y, fp_mul(zi2, zi));
    }
    memcpy(&table[w * 255], norm, sizeof(g1a) * 255);
    base = norm[255];                     /* 256 * base */
  }
  const fr tau = fr_ld(tau_m), gs = fr_ld(g_scalar_m);
  const u64 chunk = 1 << 10;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (u64 c0 = 0; c0 < n; c0 += chunk) {
    /* the chunk's points stay Jacobian until ONE shared inversion normalises them (Montgomery's trick: prefix products of
     * the z, one fp_inv — a 381-bit exponentiation, until round 6 one per point and most of this function's time —
     * then two products per point on the way back).  Same affine coordinates: they are unique. */
    g1j* acc = (g1j*)malloc(sizeof(g1j) * chunk);
    fp* pre = (fp*)malloc(sizeof(fp) * chunk);
    const u64 cnt = c0 + chunk < n ? chunk : n - c0;
    fr k = fr_mul(gs, fr_pow(tau, c0));
    fp run; memcpy(run.l, FP_ONE, 48);
    for (u64 j = 0; j < cnt; ++j) {
      uint8_t kb[32];
      fr_to_bytes(k, kb);
      g1j a = j_identity();
      for (int w = 0; w < 32; ++w) if (kb[w]) a = j_add_mixed(a, &table[w * 255 + kb[w] - 1]);
      acc[j] = a;
      pre[j] = run;                                          /* product of the finite z before j */
      if (!fp_is_zero(a.z)) run = fp_mul(run, a.z);
      k = fr_mul(k, tau);
    }
    fp inv = fp_inv(run);                                    /* 1 / (product of all finite z) */
    for (u64 j = cnt; j-- > 0;) {
      uint8_t* o = out96 + 96 * (c0 + j);
      if (fp_is_zero(acc[j].z)) { memset(o, 0, 96); continue; }   /* only for tau or g == 0 */
      const fp zi = fp_mul(inv, pre[j]), zi2 = fp_mul(zi, zi);
      inv = fp_mul(inv, acc[j].z);
      const fp x = fp_mul(acc[j].x, zi2), y = fp_mul(acc[j].y, fp_mul(zi2, zi));
      memcpy(o, x.l, 48); memcpy(o + 48, y.l, 48);
    }
    free(acc); free(pre);
  }
  free(table);
  return 0;
}
