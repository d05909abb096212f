/* CPU restatement (plain C) of the dusk-plonk prover hot path — TEST INFRASTRUCTURE ONLY.
 *
 * Used by tests/ as the larger-size checker and by bench.py's `cpu_baseline` leg.  The
 * product (plonk_amd/) never links or calls this file.
 *
 * Restates, with 64-bit-limb Montgomery arithmetic and OpenMP standing in for rayon:
 *   oracle_ntt      EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}: reference
 *                   src/fft/domain.rs:166-232, best_fft :383-422 (bit-reverse + log n
 *                   DIT stages, per-chunk parallelism, intra-chunk split for the last
 *                   stages :407-413,492-516), serial_fft :443-463, butterfly :472-489
 *   oracle_msm      msm_variable_base as called by CommitKey::commit (key.rs:376-388).
 *                   The dependency (dusk-bls12_381 0.14, not vendored) ships the
 *                   zexe/arkworks Pippenger: c = 3 if m < 32 else floor(log2 m * 69/100) + 2,
 *                   255-bit scalars, windows in parallel, running-sum bucket reduction —
 *                   restated here; the result is algorithm-independent.
 * Pinning: checked bit-for-bit against the big-int oracle (oracle/ *.py), which reproduces
 * the reference KAT digest (tests/test_oracle_c.py, tests/test_oracle_kat.py).
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ Fr (4 x 64) */
static const u64 FR_MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const u64 FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const u64 FR_R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull};
static const u64 FR_INV = 0xfffffffeffffffffull;
static const u64 FR_GEN[4] = {0x0000000efffffff1ull, 0x17e363d300189c0full, 0xff9c57876f8457b0ull, 0x351332208fc5a8c4ull};   /* 7 */
static const u64 FR_ROOT[4] = {0xb9b58d8c5f0e466aull, 0x5b1b4c801819d7ecull, 0x0af53ae352a31e64ull, 0x5bf3adda19e9b27bull}; /* 7^((q-1)/2^32) */

typedef struct { u64 l[4]; } fr;

static inline void fr_reduce(fr* r, u64 carry) {
  u64 d[4]; u128 b = 0;
  for (int i = 0; i < 4; ++i) { u128 t = (u128)r->l[i] - FR_MOD[i] - (u64)b; d[i] = (u64)t; b = (t >> 127) & 1; }
  if (!b || carry) memcpy(r->l, d, 32);
}
static inline fr fr_add(fr a, fr b) {
  fr r; u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (u64)c; c >>= 64; }
  fr_reduce(&r, (u64)c);
  return r;
}
static inline fr fr_sub(fr a, fr b) {
  fr r; u128 bw = 0;
  for (int i = 0; i < 4; ++i) { u128 t = (u128)a.l[i] - b.l[i] - (u64)bw; r.l[i] = (u64)t; bw = (t >> 127) & 1; }
  if (bw) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + FR_MOD[i]; r.l[i] = (u64)c; c >>= 64; } }
  return r;
}
static inline fr fr_mul(fr a, fr b) {
  u64 t[6] = {0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (u64)c; c >>= 64; }
    c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
    u64 m = t[0] * FR_INV;
    c = (u128)m * FR_MOD[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * FR_MOD[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
  }
  fr r; memcpy(r.l, t, 32);
  fr_reduce(&r, t[4]);
  return r;
}
static fr fr_pow(fr a, u64 e) {
  fr acc; memcpy(acc.l, FR_ONE, 32);
  for (int b = 63; b >= 0; --b) { acc = fr_mul(acc, acc); if ((e >> b) & 1) acc = fr_mul(acc, a); }
  return acc;
}
static fr fr_inv(fr a) {   /* a^(q-2) */
  u64 e[4]; memcpy(e, FR_MOD, 32); e[0] -= 2;
  fr acc; memcpy(acc.l, FR_ONE, 32);
  for (int w = 3; w >= 0; --w) for (int b = 63; b >= 0; --b) { acc = fr_mul(acc, acc); if ((e[w] >> b) & 1) acc = fr_mul(acc, a); }
  return acc;
}
static fr fr_from_u64(u64 v) { fr x = {{v, 0, 0, 0}}, r2; memcpy(r2.l, FR_R2, 32); return fr_mul(x, r2); }

/* ------------------------------------------------------------------ NTT */
static inline uint32_t bitrev(uint32_t n, uint32_t l) {   /* the low l bits of n reversed (l <= 32) */
  if (l == 0) return 0;
  n = ((n >> 1) & 0x55555555u) | ((n & 0x55555555u) << 1);
  n = ((n >> 2) & 0x33333333u) | ((n & 0x33333333u) << 2);
  n = ((n >> 4) & 0x0f0f0f0fu) | ((n & 0x0f0f0f0fu) << 4);
  n = ((n >> 8) & 0x00ff00ffu) | ((n & 0x00ff00ffu) << 8);
  n = (n >> 16) | (n << 16);
  return n >> (32 - l);
}

/* best_fft (domain.rs:383-422) with OpenMP in place of rayon: bit-reverse, then log n DIT
 * stages.  Each stage's n/2 butterflies are cut into contiguous ranges (one per task); a
 * range that starts inside a chunk seeds its running twiddle with w_m^j exactly like
 * parallel_butterfly_chunk (:492-516), and restarts from 1 at every chunk boundary like
 * butterfly_chunk (:466-469).  Below 2^12 elements it is the serial_fft (:388,443-463). */
static void best_fft(fr* a, fr omega, uint32_t log_n, int threads) {
  const u64 n = 1ull << log_n;
  if (n < (1u << 12)) threads = 1;
  /* bitreverse_permute (:429-437; serial in the reference).  Every pair (k, rk) is swapped by the one iteration with k < rk, so
   * the iterations are independent: run them on all threads (round 6: at 2^25 elements the serial loop — 33 M cache-missing
   * swaps — was a third of a transform and the 2^22-gate parity test is 22 such transforms).  Same permutation. */
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
  for (u64 k = 0; k < n; ++k) { u64 rk = bitrev((uint32_t)k, log_n); if (k < rk) { fr t = a[k]; a[k] = a[rk]; a[rk] = t; } }
  const u64 half = n / 2;
  u64 ntasks = (u64)threads * 4;
  if (ntasks > half) ntasks = half ? half : 1;
  u64 m = 1;
  for (uint32_t s = 0; s < log_n; ++s) {
    const fr w_m = fr_pow(omega, n / (2 * m));
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (u64 t = 0; t < ntasks; ++t) {
      const u64 b0 = half * t / ntasks, b1 = half * (t + 1) / ntasks;
      u64 j = b0 & (m - 1);
      fr w = j ? fr_pow(w_m, j) : *(const fr*)FR_ONE;
      for (u64 b = b0; b < b1; ++b) {
        const u64 lo = ((b >> s) << (s + 1)) | j;
        fr* L = a + lo; fr* H = L + m;
        fr tt = fr_mul(*H, w);
        *H = fr_sub(*L, tt);
        *L = fr_add(*L, tt);
        if (++j == m) { j = 0; memcpy(w.l, FR_ONE, 32); } else w = fr_mul(w, w_m);
      }
    }
    m *= 2;
  }
}

/* a: n = 1 << log_n elements in place; entries >= in_len are treated as zero. */
int oracle_ntt(u64* data, uint32_t log_n, int inverse, int coset, u64 in_len, int threads) {
  if (log_n >= 32) return -1;
  if (threads <= 0) threads = omp_get_max_threads();
  fr* a = (fr*)data;
  const u64 n = 1ull << log_n;
  if (in_len > n) in_len = n;
  for (u64 i = in_len; i < n; ++i) memset(&a[i], 0, 32);
  fr g; memcpy(g.l, FR_GEN, 32);
  fr omega; memcpy(omega.l, FR_ROOT, 32);
  for (uint32_t i = log_n; i < 32; ++i) omega = fr_mul(omega, omega);
  /* multiply a[i] by base^i for i < cnt: distribute_powers (:198-204, a serial running product in the reference) cut into
   * ranges that seed their running power with base^start — the same products, on all threads */
#define SCALE_BY_POWERS(base, cnt)                                                                   \
  do {                                                                                               \
    const u64 cnt_ = (cnt), blk_ = 1u << 14, nblk_ = (cnt_ + blk_ - 1) / blk_;                       \
    const fr base_ = (base);                                                                         \
    _Pragma("omp parallel for num_threads(threads) schedule(static) if (cnt_ >= (1u << 15))")        \
    for (u64 b_ = 0; b_ < nblk_; ++b_) {                                                             \
      const u64 lo_ = b_ * blk_, hi_ = lo_ + blk_ < cnt_ ? lo_ + blk_ : cnt_;                        \
      fr p_ = fr_pow(base_, lo_);                                                                    \
      for (u64 i_ = lo_; i_ < hi_; ++i_) { a[i_] = fr_mul(a[i_], p_); p_ = fr_mul(p_, base_); }      \
    }                                                                                                \
  } while (0)
  if (!inverse && in_len <= 2) {
    /* A polynomial of at most two coefficients (the key's constant / linear / identically-zero polynomials, the `linear`
     * evaluations X, compiler.rs:312-377): its evaluations are a0 + a1 * s * omega^i (s = 7 on the coset) — the transform
     * of a two-term input written out, not 2^25-point butterflies over zeros.  The reference runs the full transform; the
     * result is the same unique linear map (closed forms of domain.rs:620-651, tests/test_oracle_c.py). */
    const fr a0 = in_len ? a[0] : fr_from_u64(0);
    fr a1 = in_len == 2 ? a[1] : fr_from_u64(0);
    if (coset) a1 = fr_mul(a1, g);
    const u64 blk = 1u << 14, nblk = (n + blk - 1) / blk;
#pragma omp parallel for num_threads(threads) schedule(static) if (n >= (1u << 15))
    for (u64 b = 0; b < nblk; ++b) {
      const u64 lo = b * blk, hi = lo + blk < n ? lo + blk : n;
      fr t = fr_mul(a1, fr_pow(omega, lo));
      for (u64 i = lo; i < hi; ++i) { a[i] = fr_add(a0, t); t = fr_mul(t, omega); }
    }
    return 0;
  }
  if (coset && !inverse) SCALE_BY_POWERS(g, in_len);
  if (inverse) omega = fr_inv(omega);
  best_fft(a, omega, log_n, threads);
  if (inverse) {
    const fr ninv = fr_inv(fr_from_u64(n));
#pragma omp parallel for num_threads(threads) schedule(static) if (n >= (1u << 12))
    for (u64 i = 0; i < n; ++i) a[i] = fr_mul(a[i], ninv);
    if (coset) SCALE_BY_POWERS(fr_inv(g), n);
  }
#undef SCALE_BY_POWERS
  return 0;
}

/* ------------------------------------------------------------------ Fp (6 x 64) */
static const u64 FP_MOD[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 FP_ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
static const u64 FP_INV = 0x89f3fffcfffcfffdull;
typedef struct { u64 l[6]; } fp;

static inline void fp_reduce(fp* r, u64 carry) {
  u64 d[6]; u128 b = 0;
  for (int i = 0; i < 6; ++i) { u128 t = (u128)r->l[i] - FP_MOD[i] - (u64)b; d[i] = (u64)t; b = (t >> 127) & 1; }
  if (!b || carry) memcpy(r->l, d, 48);
}
static inline fp fp_add(fp a, fp b) {
  fp r; u128 c = 0;
  for (int i = 0; i < 6; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (u64)c; c >>= 64; }
  fp_reduce(&r, (u64)c);
  return r;
}
static inline fp fp_sub(fp a, fp b) {
  fp r; u128 bw = 0;
  for (int i = 0; i < 6; ++i) { u128 t = (u128)a.l[i] - b.l[i] - (u64)bw; r.l[i] = (u64)t; bw = (t >> 127) & 1; }
  if (bw) { u128 c = 0; for (int i = 0; i < 6; ++i) { c += (u128)r.l[i] + FP_MOD[i]; r.l[i] = (u64)c; c >>= 64; } }
  return r;
}
static inline fp fp_mul(fp a, fp b) {
  u64 t[8] = {0};
  for (int i = 0; i < 6; ++i) {
    u128 c = 0;
    for (int j = 0; j < 6; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (u64)c; c >>= 64; }
    c += t[6]; t[6] = (u64)c; t[7] = (u64)(c >> 64);
    u64 m = t[0] * FP_INV;
    c = (u128)m * FP_MOD[0] + t[0]; c >>= 64;
    for (int j = 1; j < 6; ++j) { c += (u128)m * FP_MOD[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t[6]; t[5] = (u64)c; t[6] = t[7] + (u64)(c >> 64);
  }
  fp r; memcpy(r.l, t, 48);
  fp_reduce(&r, t[6]);
  return r;
}
static inline int fp_is_zero(fp a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
static fp fp_inv(fp a) {
  u64 e[6]; memcpy(e, FP_MOD, 48); e[0] -= 2;
  fp acc; memcpy(acc.l, FP_ONE, 48);
  for (int w = 5; w >= 0; --w) for (int b = 63; b >= 0; --b) { acc = fp_mul(acc, acc); if ((e[w] >> b) & 1) acc = fp_mul(acc, a); }
  return acc;
}

/* ------------------------------------------------------------------ G1 Jacobian */
typedef struct { fp x, y; } g1a;
typedef struct { fp x, y, z; } g1j;   /* z == 0 : identity */

static inline g1j j_identity(void) { g1j r; memcpy(r.x.l, FP_ONE, 48); memcpy(r.y.l, FP_ONE, 48); memset(r.z.l, 0, 48); return r; }
static g1j j_double(g1j p) {
  if (fp_is_zero(p.z) || fp_is_zero(p.y)) return j_identity();
  fp A = fp_mul(p.x, p.x), B = fp_mul(p.y, p.y), C = fp_mul(B, B);
  fp xb = fp_add(p.x, B);
  fp D = fp_sub(fp_sub(fp_mul(xb, xb), A), C); D = fp_add(D, D);
  fp E = fp_add(fp_add(A, A), A), F = fp_mul(E, E);
  g1j r;
  r.x = fp_sub(F, fp_add(D, D));
  fp C8 = fp_add(C, C); C8 = fp_add(C8, C8); C8 = fp_add(C8, C8);
  r.y = fp_sub(fp_mul(E, fp_sub(D, r.x)), C8);
  fp yz = fp_mul(p.y, p.z); r.z = fp_add(yz, yz);
  return r;
}
static g1j j_add_mixed(g1j p, const g1a* q) {
  if (fp_is_zero(p.z)) { g1j r; r.x = q->x; r.y = q->y; memcpy(r.z.l, FP_ONE, 48); return r; }
  fp Z1Z1 = fp_mul(p.z, p.z), U2 = fp_mul(q->x, Z1Z1), S2 = fp_mul(fp_mul(q->y, p.z), Z1Z1);
  fp H = fp_sub(U2, p.x), rr = fp_sub(S2, p.y);
  if (fp_is_zero(H)) { if (fp_is_zero(rr)) return j_double(p); return j_identity(); }
  fp HH = fp_mul(H, H), HHH = fp_mul(H, HH), V = fp_mul(p.x, HH);
  g1j r;
  r.x = fp_sub(fp_sub(fp_mul(rr, rr), HHH), fp_add(V, V));
  r.y = fp_sub(fp_mul(rr, fp_sub(V, r.x)), fp_mul(p.y, HHH));
  r.z = fp_mul(p.z, H);
  return r;
}
static g1j j_add(g1j p, g1j q) {
  if (fp_is_zero(p.z)) return q;
  if (fp_is_zero(q.z)) return p;
  fp Z1Z1 = fp_mul(p.z, p.z), Z2Z2 = fp_mul(q.z, q.z);
  fp U1 = fp_mul(p.x, Z2Z2), U2 = fp_mul(q.x, Z1Z1);
  fp S1 = fp_mul(fp_mul(p.y, q.z), Z2Z2), S2 = fp_mul(fp_mul(q.y, p.z), Z1Z1);
  fp H = fp_sub(U2, U1), rr = fp_sub(S2, S1);
  if (fp_is_zero(H)) { if (fp_is_zero(rr)) return j_double(p); return j_identity(); }
  fp HH = fp_mul(H, H), HHH = fp_mul(H, HH), V = fp_mul(U1, HH);
  g1j r;
  r.x = fp_sub(fp_sub(fp_mul(rr, rr), HHH), fp_add(V, V));
  r.y = fp_sub(fp_mul(rr, fp_sub(V, r.x)), fp_mul(S1, HHH));
  r.z = fp_mul(fp_mul(p.z, q.z), H);
  return r;
}

/* points: m x 96 B (x || y Montgomery); scalars: m x 32 B Fr Montgomery; out: 96 B + inf flag */
int oracle_msm(const uint8_t* points, const u64* scalars, u64 m, uint8_t out[97], int threads) {
  if (threads <= 0) threads = omp_get_max_threads();
  memset(out, 0, 97);
  if (m == 0) { out[96] = 1; return 0; }
  const g1a* pts = (const g1a*)points;
  /* canonical scalars */
  u64* can = (u64*)malloc(32 * m);
  fr one_raw = {{1, 0, 0, 0}};
#pragma omp parallel for num_threads(threads) schedule(static)
  for (u64 i = 0; i < m; ++i) { fr s; memcpy(s.l, scalars + 4 * i, 32); s = fr_mul(s, one_raw); memcpy(can + 4 * i, s.l, 32); }
  uint32_t lg = 0; while ((2ull << lg) <= m) ++lg;               /* floor(log2 m) */
  const uint32_t c = m < 32 ? 3 : (lg * 69 / 100) + 2;
  const uint32_t nwin = (255 + c - 1) / c;
  g1j* wsum = (g1j*)malloc(sizeof(g1j) * nwin);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (uint32_t w = 0; w < nwin; ++w) {
    const u64 nb = (1ull << c) - 1;
    g1j* buckets = (g1j*)malloc(sizeof(g1j) * nb);
    for (u64 b = 0; b < nb; ++b) buckets[b] = j_identity();
    const uint32_t bit = w * c;
    for (u64 i = 0; i < m; ++i) {
      const u64* s = can + 4 * i;
      const uint32_t limb = bit / 64, off = bit % 64;
      u64 d = s[limb] >> off;
      if (off + c > 64 && limb + 1 < 4) d |= s[limb + 1] << (64 - off);
      d &= nb;
      if (d) buckets[d - 1] = j_add_mixed(buckets[d - 1], &pts[i]);
    }
    g1j run = j_identity(), acc = j_identity();
    for (u64 b = nb; b-- > 0;) { run = j_add(run, buckets[b]); acc = j_add(acc, run); }
    wsum[w] = acc;
    free(buckets);
  }
  g1j total = j_identity();
  for (uint32_t w = nwin; w-- > 0;) {
    for (uint32_t k = 0; k < c; ++k) total = j_double(total);
    total = j_add(total, wsum[w]);
  }
  free(wsum); free(can);
  if (fp_is_zero(total.z)) { out[96] = 1; return 0; }
  fp zi = fp_inv(total.z), zi2 = fp_mul(zi, zi);
  fp x = fp_mul(total.x, zi2), y = fp_mul(total.y, fp_mul(zi2, zi));
  memcpy(out, x.l, 48); memcpy(out + 48, y.l, 48);
  return 0;
}

int oracle_max_threads(void) { return omp_get_max_threads(); }
