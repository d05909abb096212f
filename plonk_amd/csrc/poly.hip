// Device-resident polynomial kernels around the NTT/MSM hot path:
// everything Prover::prove does between the transforms and the commitments, so that the
// 8n-sized arrays never leave HBM.
//
//   quotient_kernel        quotient_poly::compute point-wise part, reference
//                          src/proof_system/quotient_poly.rs:160-310 + every widget's
//                          compute_quotient_i (src/proof_system/widget/**/proverkey.rs)
//   perm_ratio / scans     Permutation::compute_permutation_vec, src/composer/permutation.rs:213-294
//   batch_inverse_kernel   util::batch_inversion, src/util.rs:87-117 (zeros stay zero)
//   eval_kernel            Polynomial::evaluate, src/fft/polynomial.rs:120-137
//   lincomb_kernel         linearization_poly::compute + compute_aggregate_witness sums,
//                          src/proof_system/linearization_poly.rs:168-264, key.rs:394-414
//   ruffini_*              Polynomial::ruffini, src/fft/polynomial.rs:345-367
// All arithmetic is exact Fr; results are the same field elements as the reference's.
#include <cstdlib>

#include "plonk_internal.hpp"
#include "fr29.cuh"
#include "fp_safegcd.cuh"
#include "poly.hpp"

namespace plonk {

__device__ __forceinline__ Fr ldf(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void stf(Fr* p, const Fr& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fr small(uint32_t v, const Fr& one) {   // v * 1 in Montgomery form, v <= 128
  Fr acc = Fr::zero();
  Fr p = one;
  for (uint32_t b = v; b; b >>= 1) {
    if (b & 1) acc = acc + p;
    p = p.dbl();
  }
  return acc;
}

// ---------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------
__global__ void fill_zero_kernel(Fr* p, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stf(p + i, Fr::zero());
}

// blind_poly_with_blinders (prover.rs:139-152): c[i] -= b_i ; c[n + i] = b_i
__global__ void blind_kernel(Fr* coeffs, uint64_t n, BlindArgs a) {
  const int i = threadIdx.x;
  if (i < a.count) {
    stf(coeffs + i, ldf(coeffs + i) - a.b[i]);
    stf(coeffs + n + i, a.b[i]);
  }
}

// quotient split blinding (prover.rs:547-574): t (8n coeffs) -> t_low, t_mid, t_high copied out
// with stride np (+ b_k X^n, - b_{k-1}); t_fourth stays in place at t + 3n with t[3n] -= b14.
__global__ void split_t_kernel(Fr* __restrict__ t, uint64_t n, uint64_t np, Fr* __restrict__ out, SplitArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int part = blockIdx.y;   // 0..2
  if (i >= np) return;
  Fr v = Fr::zero();
  if (i < n) v = ldf(t + (uint64_t)part * n + i);
  else if (i == n) v = a.b[part];                   // t_low += b12 X^n, t_mid += b13 X^n, t_high += b14 X^n
  if (part >= 1 && i == 0) v = v - a.b[part - 1];   // t_mid -= b12, t_high -= b13
  stf(out + (uint64_t)part * np + i, v);
  if (part == 0 && i == n + 1) stf(t + 3 * n, ldf(t + 3 * n) - a.b[2]);   // t_fourth -= b14 (nobody reads t[3n])
}

// highest index with a non-zero coefficient + 1 (Polynomial::from_coefficients_vec trim, polynomial.rs:79).
// One atomic per workgroup (a per-element atomicMax serialises ~4n+7 updates on one address).
__global__ void __launch_bounds__(256) trimmed_len_kernel(const Fr* __restrict__ p, uint64_t n, unsigned long long* out) {
  __shared__ unsigned long long sh[256];
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = 0;
  if (i < n && !ldf(p + i).is_zero()) v = i + 1;
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) { const unsigned long long o = sh[threadIdx.x + d]; if (o > sh[threadIdx.x]) sh[threadIdx.x] = o; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && sh[0]) atomicMax(out, sh[0]);
}

// ---------------------------------------------------------------------------
// Compiler::preprocess pieces (prover.hip plonk_compile)
// ---------------------------------------------------------------------------
// Permutation::compute_permutation_lagrange (src/composer/permutation.rs:141-175): the packed position
// (column << 30 | row, permutation.hpp) that wire (col, i) maps to becomes K_column * w^row; `roots` holds
// w^row for row < n.  out has one array of n values per column, `stride` elements apart.
__global__ void __launch_bounds__(256) sigma_evals_kernel(const uint32_t* __restrict__ map, const Fr* __restrict__ roots,
                                                          Fr* __restrict__ out, uint64_t n, uint64_t stride, SigmaArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t col = blockIdx.y;
  if (i >= n) return;
  const uint32_t m = map[(uint64_t)col * n + i];
  const uint32_t to = m >> 30;
  const Fr r = ldf(roots + (m & 0x3FFFFFFFu));
  stf(out + (uint64_t)col * stride + i, to ? r * a.ks[to] : r);
}
// Wire columns of a proof from the witness values (prover.rs:446-460): column `col`, row i takes the value
// of witness idx[col][i]; rows past the last gate are zero.
__global__ void __launch_bounds__(256) gather_wires_kernel(const uint32_t* __restrict__ idx, const Fr* __restrict__ values,
                                                           Fr* __restrict__ wires, uint64_t constraints, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t col = blockIdx.y;
  if (i >= n) return;
  stf(wires + (uint64_t)col * n + i, i < constraints ? ldf(values + idx[(uint64_t)col * constraints + i]) : Fr::zero());
}

__global__ void scatter_pi_kernel(Fr* dense, const uint64_t* idx, const Fr* val, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) stf(dense + idx[i], ldf(val + i));
}

// ---------------------------------------------------------------------------
// Reduced-radix helpers for the O(n) kernels below.  Values in "twiddle form" x * R'' (fr29.cuh)
// are closed under Fr29::mul (x R'' * y R'' / R'' = xy R''), so chains of data x data products
// run there: one product converts in (twiddle_from_fr), one converts out (* R / R'').
// ---------------------------------------------------------------------------
struct Tw {
  uint32_t w[9];
};
__device__ __forceinline__ Fr29 tw29(const Tw& t) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = t.w[i];
  return r;
}
static Tw tw_of(const Fr& x) {   // host: x (R form) -> x * R''
  const Fr29 t = Fr29::twiddle_from_fr(x);
  Tw r;
  for (int i = 0; i < 9; ++i) r.w[i] = t.l[i];
  return r;
}
static Tw tw_plain(const Fr& x) {   // host: re-sliced R-form value (multiplying a twiddle-form value by it converts back)
  const Fr29 t = Fr29::from_fr(x);
  Tw r;
  for (int i = 0; i < 9; ++i) r.w[i] = t.l[i];
  return r;
}
__device__ __forceinline__ Fr29 ld29_(const Fr* p) { return Fr29::from_fr(ldf(p)); }
// x^e in twiddle form (e > 0 handled by square-and-multiply; e == 0 -> one_t)
__device__ __forceinline__ Fr29 pow_tw(const Fr29& xt, uint64_t e, const Fr29& one_t) {
  Fr29 acc = one_t;
  bool started = false;
  for (int b = 63; b >= 0; --b) {
    if (started) acc = Fr29::mul(acc, acc);
    if ((e >> b) & 1) {
      acc = started ? Fr29::mul(acc, xt) : xt;
      started = true;
    }
  }
  return acc;
}

// ---------------------------------------------------------------------------
// batch inversion (util.rs:87-117, zeros skipped): one Fermat inversion per workgroup of
// 256 x 16 elements.  Every element is converted to twiddle form; per-lane prefix products,
// Hillis-Steele prefix AND suffix scans of the lane totals in LDS (limb-planar), one lane inverts
// the workgroup total, and  1/x = (1/total) * (product of everything before) * (product of
// everything after).  ~7 reduced-radix products per element + 380 per workgroup, against
// 27 32-bit-limb products per element for one inversion per 16 elements.
// ---------------------------------------------------------------------------
// The geometry is a template parameter (round 4): T lanes x E elements per workgroup.  256 x 16 (one workgroup per CU at 2^20
// elements) is the round 1-3 kernel; smaller workgroups give small arrays more of the chip and a shorter critical path (the
// inversion of ONE lane is on it either way) — poly_batch_inverse picks by size.
struct BatchInvArgs {
  Tw one_t;      // 1 * R''
  Tw one_r;      // R (plain): twiddle form -> data form
  Tw conv;       // R''^2 / R: data form -> twiddle form (Fr29::twiddle_from_fr's constant)
};
template <int BI_T>
__device__ __forceinline__ void lds_put(uint32_t (*sh)[BI_T], int t, const Fr29& v) {
#pragma unroll
  for (int i = 0; i < 9; ++i) sh[i][t] = v.l[i];
}
template <int BI_T>
__device__ __forceinline__ Fr29 lds_get(uint32_t (*sh)[BI_T], int t) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = sh[i][t];
  return r;
}
template <bool TW_IO, int BI_T, int BI_E>   // TW_IO: the array holds twiddle-form values (raw integers x * 2^261 mod q), in and out
__global__ void __launch_bounds__(BI_T) batch_inverse_kernel(Fr* __restrict__ v, uint64_t n, BatchInvArgs a) {
  __shared__ uint32_t shp[9][BI_T], shs[9][BI_T], shinv[9];
  const int t = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * (BI_T * BI_E) + t;   // element k of this lane: base + k * BI_T (coalesced)
  const Fr29 one_t = tw29(a.one_t), conv = tw29(a.conv);
  Fr29 pre[BI_E];
  Fr29 acc = one_t;
  uint32_t nz = 0;
#pragma unroll
  for (int k = 0; k < BI_E; ++k) {
    pre[k] = acc;
    const uint64_t i = base + (uint64_t)k * BI_T;
    if (i < n) {
      const Fr x = ldf(v + i);
      if (!x.is_zero()) {
        nz |= 1u << k;
        acc = Fr29::mul(acc, TW_IO ? Fr29::from_fr(x) : Fr29::mul(Fr29::from_fr(x), conv).csub_q());
      }
    }
  }
  // exclusive prefix (shp) and suffix (shs) products of the lane totals
  Fr29 pfx = acc, sfx = acc;
  for (int d = 1; d < BI_T; d <<= 1) {
    lds_put(shp, t, pfx);
    lds_put(shs, t, sfx);
    __syncthreads();
    if (t >= d) pfx = Fr29::mul(pfx, lds_get(shp, t - d));
    if (t + d < BI_T) sfx = Fr29::mul(sfx, lds_get(shs, t + d));
    __syncthreads();
  }
  // pfx = prod_{s <= t} total_s, sfx = prod_{s >= t} total_s
  lds_put(shp, t, pfx);
  lds_put(shs, t, sfx);
  __syncthreads();
  if (t == 0) {   // 1 / (product of the whole workgroup): safegcd division steps (fp_safegcd.cuh) — ~13 k instructions on this
                  // one lane while 255 wait, against ~65 k for the Fermat chain sfx^(q-2) it replaces (r03)
    const Fr29 r = fr29_inv_gcd_tw(sfx);
#pragma unroll
    for (int i = 0; i < 9; ++i) shinv[i] = r.l[i];
  }
  __syncthreads();
  Fr29 outer;   // (1/total) * prod_{s < t} total_s * prod_{s > t} total_s
#pragma unroll
  for (int i = 0; i < 9; ++i) outer.l[i] = shinv[i];
  if (t > 0) outer = Fr29::mul(outer, lds_get(shp, t - 1));
  if (t + 1 < BI_T) outer = Fr29::mul(outer, lds_get(shs, t + 1));
  // within the lane: 1/x_k = outer * pre_k * (product of the lane's later elements)
  const Fr29 one_r = tw29(a.one_r);
  Fr29 suf = outer;
#pragma unroll
  for (int k = BI_E - 1; k >= 0; --k) {
    const uint64_t i = base + (uint64_t)k * BI_T;
    if ((nz >> k) & 1) {
      const Fr29 xt = TW_IO ? Fr29::from_fr(ldf(v + i)) : Fr29::mul(Fr29::from_fr(ldf(v + i)), conv).csub_q();
      const Fr29 inv_t = Fr29::mul(suf, pre[k]);
      stf(v + i, TW_IO ? inv_t.to_fr() : Fr29::mul(inv_t, one_r).to_fr());
      suf = Fr29::mul(suf, xt);
    }
  }
}

// ---------------------------------------------------------------------------
// scans over Fr with op = mul (prefix products) or add (suffix sums), 3 phases
// ---------------------------------------------------------------------------
static constexpr int SCAN_T = 256;
static constexpr int SCAN_E = 8;
static constexpr int SCAN_BLOCK = SCAN_T * SCAN_E;

template <bool MUL>
__device__ __forceinline__ Fr sop(const Fr& a, const Fr& b) {
  if constexpr (MUL) return a * b; else return a + b;
}
template <bool MUL>
__device__ __forceinline__ Fr sid() {
  if constexpr (MUL) return Fr::one(); else return Fr::zero();
}

// inclusive scan within blocks of 2048; REV scans from the high index down.
template <bool MUL, bool REV>
__global__ void __launch_bounds__(SCAN_T) scan_block_kernel(Fr* __restrict__ data, uint64_t n, Fr* __restrict__ totals) {
  __shared__ Fr sh[SCAN_T];
  const int t = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)t * SCAN_E;
  Fr v[SCAN_E];
  Fr acc = sid<MUL>();
#pragma unroll
  for (int k = 0; k < SCAN_E; ++k) {
    const uint64_t li = base + k;
    if (li < n) {
      const uint64_t gi = REV ? (n - 1 - li) : li;
      acc = sop<MUL>(acc, ldf(data + gi));
    }
    v[k] = acc;
  }
  sh[t] = acc;
  __syncthreads();
  for (int d = 1; d < SCAN_T; d <<= 1) {
    Fr x = sid<MUL>();
    const bool take = t >= d;
    if (take) x = sh[t - d];
    __syncthreads();
    if (take) sh[t] = sop<MUL>(x, sh[t]);
    __syncthreads();
  }
  const Fr excl = t ? sh[t - 1] : sid<MUL>();
  if (t == SCAN_T - 1) stf(totals + blockIdx.x, sh[t]);
#pragma unroll
  for (int k = 0; k < SCAN_E; ++k) {
    const uint64_t li = base + k;
    if (li < n) {
      const uint64_t gi = REV ? (n - 1 - li) : li;
      stf(data + gi, sop<MUL>(excl, v[k]));
    }
  }
}

// single-block inclusive scan of the block totals (nb <= SCAN_T * SCAN_MAXPER); each thread owns
// a contiguous run of `per` totals.
static constexpr int SCAN_MAXPER = 64;
template <bool MUL>
__global__ void __launch_bounds__(SCAN_T) scan_totals_kernel(Fr* __restrict__ totals, uint32_t nb, uint32_t per) {
  __shared__ Fr sh[SCAN_T];
  const int t = threadIdx.x;
  Fr acc = sid<MUL>();
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t i = t * per + k;
    if (i < nb) acc = sop<MUL>(acc, ldf(totals + i));
  }
  sh[t] = acc;
  __syncthreads();
  for (int d = 1; d < SCAN_T; d <<= 1) {
    Fr x = sid<MUL>();
    const bool take = t >= d;
    if (take) x = sh[t - d];
    __syncthreads();
    if (take) sh[t] = sop<MUL>(x, sh[t]);
    __syncthreads();
  }
  acc = t ? sh[t - 1] : sid<MUL>();
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t i = t * per + k;
    if (i < nb) {
      acc = sop<MUL>(acc, ldf(totals + i));
      stf(totals + i, acc);
    }
  }
}

template <bool MUL, bool REV>
__global__ void __launch_bounds__(SCAN_T) scan_apply_kernel(Fr* __restrict__ data, uint64_t n, const Fr* __restrict__ totals) {
  if (blockIdx.x == 0) return;
  const Fr off = ldf(totals + blockIdx.x - 1);
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_E;
#pragma unroll
  for (int k = 0; k < SCAN_E; ++k) {
    const uint64_t li = base + k;
    if (li < n) {
      const uint64_t gi = REV ? (n - 1 - li) : li;
      stf(data + gi, sop<MUL>(off, ldf(data + gi)));
    }
  }
}

template <bool MUL, bool REV>
static int scan_inplace(Ctx* c, Fr* data, uint64_t n, Fr* totals) {
  const uint32_t nb = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  if (nb > SCAN_T * SCAN_MAXPER) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);   // n <= 2^25
  hipLaunchKernelGGL((scan_block_kernel<MUL, REV>), dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals);
  if (nb > 1) {
    hipLaunchKernelGGL((scan_totals_kernel<MUL>), dim3(1), dim3(SCAN_T), 0, c->stream, totals, nb, (nb + SCAN_T - 1) / SCAN_T);
    hipLaunchKernelGGL((scan_apply_kernel<MUL, REV>), dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals);
  }
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
// ---- prefix products of twiddle-form values (the grand product z): same three phases in Fr29, and the
// last phase converts back to the data domain (x * 2^256) for the inverse FFT that follows
struct PScanArgs {
  Tw one_t, one_r;
  Tw carry_t;
  int has_carry;
};
__device__ __forceinline__ void ps_put(uint32_t (*sh)[SCAN_T], int t, const Fr29& v) {
#pragma unroll
  for (int i = 0; i < 9; ++i) sh[i][t] = v.l[i];
}
__device__ __forceinline__ Fr29 ps_get(uint32_t (*sh)[SCAN_T], int t) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = sh[i][t];
  return r;
}
// Hillis-Steele inclusive product scan of one value per lane; returns the exclusive prefix, total in *tot
__device__ __forceinline__ Fr29 ps_lanes(Fr29 acc, uint32_t (*sh)[SCAN_T], int t, const Fr29& one_t, Fr29* tot) {
  for (int d = 1; d < SCAN_T; d <<= 1) {
    ps_put(sh, t, acc);
    __syncthreads();
    if (t >= d) acc = Fr29::mul(ps_get(sh, t - d), acc);
    __syncthreads();
  }
  ps_put(sh, t, acc);
  __syncthreads();
  const Fr29 excl = t ? ps_get(sh, t - 1) : one_t;
  *tot = ps_get(sh, SCAN_T - 1);
  return excl;
}
// Elements per lane (round 6, second session): 8 for large arrays (throughput), 2 up to 2^17 elements — below that the three
// kernels are chains of dependent products on a mostly idle chip (8 + 8 + 8 in the block pass, 2 + 8 in the apply pass:
// 27 + 20 + 21 us at 2^16 elements), and a quarter of the elements per lane is a quarter of the serial part.
static inline int pscan_e(uint64_t n) { return n <= (1ull << 17) ? 2 : 8; }
template <bool FINAL, int E>   // FINAL: single workgroup, results leave in the data domain
__global__ void __launch_bounds__(SCAN_T) pscan_block_kernel(Fr* __restrict__ data, uint64_t n, Fr* __restrict__ totals, PScanArgs a) {
  __shared__ uint32_t sh[9][SCAN_T];
  const int t = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * (SCAN_T * E) + (uint64_t)t * E;
  const Fr29 one_t = tw29(a.one_t);
  Fr29 v[E];
  Fr29 acc = one_t;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    if (base + k < n) acc = Fr29::mul(acc, ld29_(data + base + k));
    v[k] = acc;
  }
  Fr29 tot;
  const Fr29 excl = ps_lanes(acc, sh, t, one_t, &tot);
  if (t == 0) stf(totals + blockIdx.x, tot.to_fr());
  const Fr29 post = FINAL ? Fr29::mul(excl, tw29(a.one_r)) : excl;   // (x R'') * R / R'' = x R
#pragma unroll
  for (int k = 0; k < E; ++k)
    if (base + k < n) stf(data + base + k, Fr29::mul(post, v[k]).to_fr());
}
__global__ void __launch_bounds__(SCAN_T) pscan_totals_kernel(Fr* __restrict__ totals, uint32_t nb, uint32_t per, PScanArgs a) {
  __shared__ uint32_t sh[9][SCAN_T];
  const int t = threadIdx.x;
  const Fr29 one_t = tw29(a.one_t);
  Fr29 acc = one_t;
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t i = t * per + k;
    if (i < nb) acc = Fr29::mul(acc, ld29_(totals + i));
  }
  Fr29 tot;
  acc = ps_lanes(acc, sh, t, one_t, &tot);
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t i = t * per + k;
    if (i < nb) {
      acc = Fr29::mul(acc, ld29_(totals + i));
      stf(totals + i, acc.to_fr());
    }
  }
}
template <int E>
__global__ void __launch_bounds__(SCAN_T) pscan_apply_kernel(Fr* __restrict__ data, uint64_t n, const Fr* __restrict__ totals, PScanArgs a) {
  // block 0 only converts; the others also multiply by the product of everything before them
  Fr29 off = tw29(a.one_r);
  if (a.has_carry) off = Fr29::mul(tw29(a.carry_t), off);   // a range of a longer product: everything before the range
  if (blockIdx.x) off = Fr29::mul(ld29_(totals + blockIdx.x - 1), off);
  const uint64_t base = (uint64_t)blockIdx.x * (SCAN_T * E) + (uint64_t)threadIdx.x * E;
#pragma unroll
  for (int k = 0; k < E; ++k)
    if (base + k < n) stf(data + base + k, Fr29::mul(ld29_(data + base + k), off).to_fr());
}
uint32_t scan_prefix_blocks(uint64_t n) { const uint64_t blk = (uint64_t)SCAN_T * pscan_e(n); return (uint32_t)((n + blk - 1) / blk); }
static void pscan_launch_block(Ctx* c, bool final, uint32_t nb, Fr* data, uint64_t n, Fr* totals, const PScanArgs& a) {
  if (pscan_e(n) == 2) {
    if (final) hipLaunchKernelGGL((pscan_block_kernel<true, 2>), dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals, a);
    else hipLaunchKernelGGL((pscan_block_kernel<false, 2>), dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals, a);
  } else {
    if (final) hipLaunchKernelGGL((pscan_block_kernel<true, 8>), dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals, a);
    else hipLaunchKernelGGL((pscan_block_kernel<false, 8>), dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals, a);
  }
}
static void pscan_launch_apply(Ctx* c, uint32_t nb, Fr* data, uint64_t n, const Fr* totals, const PScanArgs& a) {
  if (pscan_e(n) == 2) hipLaunchKernelGGL(pscan_apply_kernel<2>, dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals, a);
  else hipLaunchKernelGGL(pscan_apply_kernel<8>, dim3(nb), dim3(SCAN_T), 0, c->stream, data, n, totals, a);
}
// data: twiddle form in, data domain (Montgomery R = 2^256) out
int scan_prefix_product(Ctx* c, Fr* data, uint64_t n, Fr* totals) {
  const uint32_t nb = scan_prefix_blocks(n);
  if (nb > SCAN_T * SCAN_MAXPER) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);   // n <= 2^25
  PScanArgs a{};
  a.one_t = tw_of(Fr::one());
  a.one_r = tw_plain(Fr::one());
  if (nb == 1) {
    pscan_launch_block(c, true, 1, data, n, totals, a);
  } else {
    pscan_launch_block(c, false, nb, data, n, totals, a);
    hipLaunchKernelGGL(pscan_totals_kernel, dim3(1), dim3(SCAN_T), 0, c->stream, totals, nb, (nb + SCAN_T - 1) / SCAN_T, a);
    pscan_launch_apply(c, nb, data, n, totals, a);
  }
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int scan_prefix_product_local(Ctx* c, Fr* data, uint64_t n, Fr* totals) {
  const uint32_t nb = scan_prefix_blocks(n);
  if (nb == 0 || nb > SCAN_T * SCAN_MAXPER) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  PScanArgs a{};
  a.one_t = tw_of(Fr::one());
  a.one_r = tw_plain(Fr::one());
  pscan_launch_block(c, false, nb, data, n, totals, a);
  hipLaunchKernelGGL(pscan_totals_kernel, dim3(1), dim3(SCAN_T), 0, c->stream, totals, nb, (nb + SCAN_T - 1) / SCAN_T, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int scan_prefix_product_apply(Ctx* c, Fr* data, uint64_t n, const Fr* totals, const Fr& carry_twiddle) {
  const uint32_t nb = scan_prefix_blocks(n);
  PScanArgs a{};
  a.one_t = tw_of(Fr::one());
  a.one_r = tw_plain(Fr::one());
  a.carry_t = tw_plain(carry_twiddle);   // already x * 2^261: re-sliced, not converted
  a.has_carry = 1;
  pscan_launch_apply(c, nb, data, n, totals, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int scan_suffix_sum(Ctx* c, Fr* data, uint64_t n, Fr* totals) { return scan_inplace<false, true>(c, data, n, totals); }

// ---------------------------------------------------------------------------
// permutation grand product inputs (permutation.rs:251-294)
//   s[0] = 1 ; s[i] = num[i-1] (numerators) ; den[i] = denominators[i-1] (inverted later)
// ---------------------------------------------------------------------------
// Output in twiddle form (raw integers x * 2^261 mod q): the products below, the batch inversion, the
// num/den product and the prefix-product scan all stay in that form; the scan converts back.
struct PermConst {
  Tw conv;       // 32 * R'': data -> twiddle form
  Tw gamma_t;    // gamma
  Tw bk_t[4];    // beta * {1, K1, K2, K3}
  Tw b32_t;      // 32 * beta: sigma (data form) * this = beta * sigma in twiddle form
  Tw one_t;
};
__device__ __forceinline__ Fr29 ld_slot(const void* base, uint64_t idx) {
  const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const Fr29Slot*>(base) + idx);
  const uint4 a = q[0], b = q[1];
  const uint32_t c = reinterpret_cast<const Fr29Slot*>(base)[idx].w[8];
  Fr29 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c;
  return r;
}
__global__ void __launch_bounds__(128) perm_terms_kernel(PermArgs a, PermConst k) {
  const uint64_t j0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j0 >= a.count) return;
  const uint64_t i = a.first + j0;
  if (i == 0) {
    const Fr one = tw29(k.one_t).to_fr();
    stf(a.num, one);
    stf(a.den, one);
    return;
  }
  const uint64_t r = i - 1;
  Fr29 root = ld_slot(a.tw_lo29, r & ((1ull << a.lobits) - 1));
  if (a.use_hi) root = Fr29::mul(root, ld_slot(a.tw_hi29, r >> a.lobits));
  const Fr29 conv = tw29(k.conv), gamma_t = tw29(k.gamma_t), b32 = tw29(k.b32_t);
  Fr29 num, den;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const Fr29 wg = Fr29::add_csub(Fr29::mul(ld29_(a.wires[j] + r), conv), gamma_t);
    const Fr29 nf = Fr29::add_csub(wg, Fr29::mul(root, tw29(k.bk_t[j])));
    const Fr29 df = Fr29::add_csub(wg, Fr29::mul(ld29_(a.sigma[j] + r), b32));
    num = j ? Fr29::mul(num, nf) : nf;
    den = j ? Fr29::mul(den, df) : df;
  }
  stf(a.num + i, num.to_fr());
  stf(a.den + i, den.to_fr());
}
__global__ void mul_arrays_kernel(Fr* __restrict__ a, const Fr* __restrict__ b, uint64_t n, int* zero_flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr y = ldf(b + i);
  if (zero_flag && y.is_zero()) *zero_flag = 1;
  stf(a + i, Fr29::mul(ld29_(a + i), Fr29::from_fr(y)).to_fr());   // twiddle form is closed under the product
}

// ---------------------------------------------------------------------------
// quotient numerator / Z_H on the 8n coset (quotient_poly.rs:96-101,160-310)
// ---------------------------------------------------------------------------
__device__ __forceinline__ Fr delta4(const Fr& f, const Fr& one) {   // f (f-1)(f-2)(f-3)
  const Fr two = one.dbl();
  return f * (f - one) * (f - two) * (f - two - one);
}

// Hot part (arithmetic + permutation + L1 + Z_H division) in the reduced-radix lazy form of
// fr29.cuh: 28 Montgomery products per point at ~220 VALU instructions each.  Data stays in the
// reference's R = 2^256 domain while the reduction is by R'' = 2^261, so every product of two
// DATA values carries an extra 2^-5; that is pre-compensated once: the selector / L1 evaluation
// arrays are stored scaled (q_m by 2^10; q_l q_r q_o q_f q_arith l1 by 2^5, prover.hip) and the
// challenge constants are handed over as c * 2^(5k) * R'' (QuotientConst).  The range / logic /
// fixed-base / curve-addition widgets — identically zero selectors in most circuits — run on
// the exact 32-bit path from the same loaded values.
__device__ __forceinline__ Fr29 ld29(const Fr* p) { return Fr29::from_fr(ldf(p)); }
__device__ __forceinline__ Fr29 c29(const uint32_t (&v)[9]) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = v[i];
  return r;
}

// WIDGETS = false: circuits without range / logic / ECC gates (their selector polynomials are
// identically zero) get a kernel without the 32-bit exact path — fewer registers, more waves in
// flight for what is the most bandwidth-hungry pass of a proof.
// twiddle-form constants of the widget path (poly_quotient fills them per proof)
struct WidgetConst {
  Tw conv, one_r, c1, c3, c9, c18, c81, c83, ed;
  Tw rg[4];   // range_ch * {1, k, k^2, k^3},      k = range_ch^2
  Tw lg[5];   // logic_ch * {k^3, 1, k, k^2, k^4}, k = logic_ch^2   (order of use in the kernel)
  Tw fx[4];   // fixed_ch * {1, k, k^2, k^3},      k = fixed_ch^2
  Tw vr[3];   // var_ch * {1, k, k^2},             k = var_ch^2
};
template <bool WIDGETS>
__global__ void __launch_bounds__(128, WIDGETS ? 2 : 4) quotient_kernel(QuotientArgs q, WidgetConst wc) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= q.n8) return;
  const uint64_t iw = (i + q.rot) & (q.n8 - 1);       // extended arrays wrap (quotient_poly.rs:61-67)
  const Fr29 a = ld29(q.a + i), b = ld29(q.b + i), c = ld29(q.c + i), d = ld29(q.d + i);
  const Fr29 z = ld29(q.z + i), z_w = ld29(q.z + iw);
  const Fr29 gamma = c29(q.k.gamma);
  Fr29 t = q.pi ? ld29(q.pi + i) : Fr29::zero();   // PI(X) = 0 when the circuit has no public inputs
  // arithmetic (arithmetic/proverkey.rs:44-71); selector arrays pre-scaled, see above
  {
    Fr29 s = ld29(q.q_c + i);
    if (q.has[QS_M]) s = Fr29::add_csub(s, Fr29::mul(Fr29::mul(a, b), ld29(q.q_m + i)));
    if (q.has[QS_L]) s = Fr29::add_csub(s, Fr29::mul(a, ld29(q.q_l + i)));
    if (q.has[QS_R]) s = Fr29::add_csub(s, Fr29::mul(b, ld29(q.q_r + i)));
    if (q.has[QS_O]) s = Fr29::add_csub(s, Fr29::mul(c, ld29(q.q_o + i)));
    if (q.has[QS_F]) s = Fr29::add_csub(s, Fr29::mul(d, ld29(q.q_f + i)));
    t = Fr29::add_csub(t, Fr29::mul(s, ld29(q.q_arith + i)));
  }
  // permutation (permutation/proverkey.rs:40-125)
  {
    const Fr29 x = ld29(q.linear + i);
    const Fr29 ag = Fr29::add_csub(a, gamma), bg = Fr29::add_csub(b, gamma);
    const Fr29 cg = Fr29::add_csub(c, gamma), dg = Fr29::add_csub(d, gamma);
    Fr29 p1 = Fr29::mul(Fr29::add_csub(ag, Fr29::mul(x, c29(q.k.beta_k[0]))),
                        Fr29::add_csub(bg, Fr29::mul(x, c29(q.k.beta_k[1]))));
    p1 = Fr29::mul(p1, Fr29::add_csub(cg, Fr29::mul(x, c29(q.k.beta_k[2]))));
    p1 = Fr29::mul(p1, Fr29::add_csub(dg, Fr29::mul(x, c29(q.k.beta_k[3]))));
    t = Fr29::add_csub(t, Fr29::mul(Fr29::mul(p1, z), c29(q.k.alpha_pos)));            // identity * z * alpha
    const Fr29 be = c29(q.k.beta_k[0]);
    Fr29 p2 = Fr29::mul(Fr29::add_csub(ag, Fr29::mul(ld29(q.s1 + i), be)), Fr29::add_csub(bg, Fr29::mul(ld29(q.s2 + i), be)));
    p2 = Fr29::mul(p2, Fr29::add_csub(cg, Fr29::mul(ld29(q.s3 + i), be)));
    p2 = Fr29::mul(p2, Fr29::add_csub(dg, Fr29::mul(ld29(q.s4 + i), be)));
    t = Fr29::add_csub(t, Fr29::mul(Fr29::mul(p2, z_w), c29(q.k.alpha_neg)));          // - copy * z_w * alpha
    const Fr29 l1a = Fr29::mul(ld29(q.l1 + i), c29(q.k.alpha_sq));                      // L1 * alpha^2 (* 2^5)
    t = Fr29::add_csub(t, Fr29::mul(Fr29::sub_lazy(z, c29(q.k.one)), l1a));             // (z - 1) L1 alpha^2
  }
  if constexpr (WIDGETS) {
    // Remaining widgets, also in reduced radix.  Their formulas are long chains of data x data
    // products, so everything is moved to "twiddle form" x * 2^261 (closed under Fr29::mul) with one
    // product per operand (the arrays stored pre-scaled by 2^5 — q_l, q_r — already ARE in that form
    // when re-sliced), the separation challenges and their powers come in as twiddle-form constants,
    // and the widget sum returns to the data domain with one product at the end.
    // Range discipline: products take at most ONE lazy operand (sub_lazy: a - b + 4q, first argument);
    // every other difference is sub_reduce'd to [0, 2q + eps).
    using F = Fr29;
    const F conv = tw29(wc.conv), c1 = tw29(wc.c1);
    auto T = [&](const Fr* p) { return F::mul(ld29(p), conv).csub_q(); };
    auto add = [](const F& x, const F& y) { return F::add_csub(x, y); };
    auto subn = [](const F& x, const F& y) { return F::sub_reduce(x, y); };
    auto subl = [](const F& x, const F& y) { return F::sub_lazy(x, y); };
    auto mul = [](const F& x, const F& y) { return F::mul(x, y); };
    auto x4 = [&](const F& x) { const F d2 = add(x, x); return add(d2, d2); };
    auto delta = [&](const F& f) {   // f (f-1)(f-2)(f-3)
      const F c2 = add(c1, c1);
      F r = mul(subl(f, c1), f);
      r = mul(subl(f, c2), r);
      return mul(subl(f, add(c2, c1)), r);
    };
    const F a_ = T(q.a + i), b_ = T(q.b + i), c_ = T(q.c + i), d_ = T(q.d + i);
    const F a_w = T(q.a + iw), b_w = T(q.b + iw), d_w = T(q.d + iw);
    F u = F::zero();
    if (q.has[QS_RANGE]) {   // range/proverkey.rs:32-58
      F s = mul(delta(subn(c_, x4(d_))), tw29(wc.rg[0]));
      s = add(s, mul(delta(subn(b_, x4(c_))), tw29(wc.rg[1])));
      s = add(s, mul(delta(subn(a_, x4(b_))), tw29(wc.rg[2])));
      s = add(s, mul(delta(subn(d_w, x4(a_))), tw29(wc.rg[3])));
      u = add(u, mul(s, T(q.q_range + i)));
    }
    if (q.has[QS_LOGIC]) {   // logic/proverkey.rs:34-70,108-144
      const F c3 = tw29(wc.c3), c18 = tw29(wc.c18), c81 = tw29(wc.c81);
      const F la = subn(a_w, x4(a_)), lb = subn(b_w, x4(b_)), ld = subn(d_w, x4(d_)), w = c_;
      const F ab = add(la, lb);
      const F in1 = subn(add(x4(w), c81), mul(ab, c18));                           // 4w - 18(a+b) + 81
      const F in2 = subn(add(add(mul(in1, w), mul(add(mul(la, la), mul(lb, lb)), c18)), tw29(wc.c83)), mul(ab, c81));
      const F Fv = mul(in2, w);
      const F Ee = subn(mul(add(ab, ld), c3), add(Fv, Fv));
      const F Bb = mul(subn(mul(ld, tw29(wc.c9)), mul(ab, c3)), T(q.q_c + i));
      F s = mul(subl(w, mul(la, lb)), tw29(wc.lg[0]));
      s = add(s, mul(delta(la), tw29(wc.lg[1])));
      s = add(s, mul(delta(lb), tw29(wc.lg[2])));
      s = add(s, mul(delta(ld), tw29(wc.lg[3])));
      s = add(s, mul(add(Bb, Ee), tw29(wc.lg[4])));
      u = add(u, mul(s, T(q.q_logic + i)));
    }
    if (q.has[QS_FIXED]) {   // ecc/scalar_mul/fixed_base/proverkey.rs:39-101
      const F x_beta = ld29(q.q_l + i), y_beta = ld29(q.q_r + i), q_c = T(q.q_c + i);   // q_l, q_r: stored * 2^5
      const F bit = subn(d_w, add(d_, d_));
      const F bit_cons = mul(mul(subl(bit, c1), bit), add(bit, c1));
      const F y_alpha = add(mul(subl(y_beta, c1), mul(bit, bit)), c1);
      const F x_alpha = mul(bit, x_beta);
      const F xy_cons = mul(subl(mul(bit, q_c), c_), tw29(wc.fx[1]));
      const F cab = mul(mul(mul(c_, a_), b_), tw29(wc.ed));
      const F x_acc = mul(subl(add(a_w, mul(a_w, cab)), add(mul(a_, y_alpha), mul(b_, x_alpha))), tw29(wc.fx[2]));
      const F y_acc = mul(subl(subn(b_w, mul(b_w, cab)), add(mul(b_, y_alpha), mul(a_, x_alpha))), tw29(wc.fx[3]));
      const F s = add(add(mul(bit_cons, tw29(wc.fx[0])), x_acc), add(y_acc, xy_cons));
      u = add(u, mul(s, T(q.q_fixed + i)));
    }
    if (q.has[QS_VAR]) {     // ecc/curve_addition/proverkey.rs:33-77
      const F x1y2 = d_w;
      const F y1x2 = mul(b_, c_), y1y2 = mul(b_, d_), x1x2 = mul(a_, c_);
      const F xy_cons = mul(subl(mul(a_, d_), x1y2), tw29(wc.vr[0]));
      const F dxy = mul(mul(x1y2, y1x2), tw29(wc.ed));
      const F x3c = mul(subl(add(x1y2, y1x2), add(a_w, mul(a_w, dxy))), tw29(wc.vr[1]));
      const F y3c = mul(subl(add(y1y2, x1x2), subn(b_w, mul(b_w, dxy))), tw29(wc.vr[2]));
      u = add(u, mul(add(add(xy_cons, x3c), y3c), T(q.q_var + i)));
    }
    t = Fr29::add_csub(t, Fr29::mul(u, tw29(wc.one_r)));   // back to the data domain
  }
  stf(q.out + i, Fr29::mul(t, c29(q.k.vinv[i & 7])).to_fr());
}

// t holds A = T mod (X^nq - g^nq) (nq coefficients).  T = T_lo + X^nq T_hi with deg T_hi <= 6, so
// A[k] = T[k] + g^nq T[nq + k] for k < 7 and A[k] = T[k] above.  Given the true low coefficients
// T[0..7) this restores T in place: t[k] = low[k], t[nq + k] = (A[k] - low[k]) / g^nq, zeros above.
struct DealiasArgs {
  Fr low[7];
  Fr g_inv;
};
__global__ void dealias_kernel(Fr* __restrict__ t, uint64_t nq, DealiasArgs a) {
  const uint32_t k = threadIdx.x;
  if (k < 7) {
    const Fr A = ldf(t + k);
    stf(t + nq + k, (A - a.low[k]) * a.g_inv);
    stf(t + k, a.low[k]);
  } else if (k < 16) {
    stf(t + nq + k, Fr::zero());
  }
}

// v[i] *= s  (one-off pre-scaling of key arrays for the kernel above)
__global__ void scale_array_kernel(Fr* __restrict__ v, uint64_t n, Fr s) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stf(v + i, ldf(v + i) * s);
}

// Montgomery limbs -> canonical little-endian scalars (BlsScalar::to_bytes), for the serialisers
__global__ void from_mont_kernel(const Fr* __restrict__ src, Fr* __restrict__ dst, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stf(dst + i, ldf(src + i).from_mont());
}

// L1 numerators on the coset: out[i] = v_h[i & 7] * n_inv  (to be multiplied by 1/(linear[i]-1))
__global__ void l1_prepare_kernel(const Fr* __restrict__ linear, Fr* __restrict__ out, uint64_t n8) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n8) stf(out + i, ldf(linear + i) - Fr::one());
}
__global__ void l1_finish_kernel(Fr* __restrict__ l1, uint64_t n8, L1Args a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n8) stf(l1 + i, ldf(l1 + i) * a.vh[i & 7] * a.n_inv);
}

// ---------------------------------------------------------------------------
// polynomial evaluation (Polynomial::evaluate, polynomial.rs:120-137): each lane runs Horner over
// 2^LE coefficients with the point in twiddle form; lanes, then workgroups, are combined by a
// tree in which the right half is multiplied by x^(distance).  The squarings x^(2^b) are computed by
// the HOST, once per distinct point of a call (two in a proof: z and z w), and ride in the kernel
// arguments (round 6, second session): until then lane 0 of every workgroup ran the chain (11, then 20
// dependent squarings) while the other 255 waited for it — with 16 coefficients per lane the two kernels
// were ~53 dependent products = 72 us at EVERY size up to 2^16 gates.  Small polynomials now take 4
// coefficients per lane (4 x the workgroups, a quarter of the Horner chain) and the final tree stops at
// the number of partials: ~20 dependent products.
// ---------------------------------------------------------------------------
static constexpr int EV_T = 256;
static constexpr int EV_PW = 22;    // x^(2^b), b < 22: 2^(8 + LE) coefficients per workgroup, 256 partials per final lane step
static constexpr int EV_POINTS = 2; // distinct evaluation points per call (a proof has two: z and z w)
struct EvalItemD {
  const Fr* poly;
  uint64_t len;
  uint32_t point;   // index into pw
};
struct EvalArgsD {
  EvalItemD items[16];
  Tw pw[EV_POINTS][EV_PW];
  Fr* partial;
  uint32_t max_blocks;
};
// tree over the first 2^levels lane values; lane t's value stands for x^(step * t) * v_t, pw[s] = x^(step * 2^s)
__device__ __forceinline__ Fr29 eval_tree(Fr29 acc, uint32_t (*sh)[EV_T], const Tw* pw, int t, int levels) {
  for (int s = 0; s < levels; ++s) {
    const int d = 1 << s;
    lds_put(sh, t, acc);
    __syncthreads();
    if ((t & (2 * d - 1)) == 0) acc = Fr29::add_csub(acc, Fr29::mul(lds_get(sh, t + d), tw29(pw[s])));
    __syncthreads();
  }
  return acc;
}
template <int LE>   // 2^LE coefficients per lane, 2^(8 + LE) per workgroup
__global__ void __launch_bounds__(EV_T) eval_kernel(EvalArgsD a) {
  constexpr int E = 1 << LE;
  __shared__ uint32_t sh[9][EV_T];
  const EvalItemD& it = a.items[blockIdx.y];
  const Tw* pw = a.pw[it.point];
  const int t = threadIdx.x;
  const Fr29 xt = tw29(pw[0]);
  const uint64_t base = ((uint64_t)blockIdx.x * EV_T + t) * E;
  Fr29 acc = Fr29::zero();
  if (base < it.len) {
#pragma unroll
    for (int k = E - 1; k >= 0; --k) {
      acc = Fr29::mul(acc, xt);
      if (base + k < it.len) acc = Fr29::add_csub(acc, ld29_(it.poly + base + k));
    }
  }
  acc = eval_tree(acc, sh, pw + LE, t, 8);
  if (t == 0) stf(a.partial + (uint64_t)blockIdx.y * a.max_blocks + blockIdx.x, acc.to_fr());
}
template <int LE>
__global__ void __launch_bounds__(EV_T) eval_final_kernel(EvalArgsD a, uint32_t nblocks, int levels, Fr* __restrict__ out) {
  __shared__ uint32_t sh[9][EV_T];
  const EvalItemD& it = a.items[blockIdx.x];
  const Tw* pw = a.pw[it.point];
  const int t = threadIdx.x;
  // lane t: Horner in x^(2^(16 + LE)) over workgroup partials t, t + 256, ...
  Fr29 acc = Fr29::zero();
  const Fr* part = a.partial + (uint64_t)blockIdx.x * a.max_blocks;
  if ((uint32_t)t < nblocks) {
    const uint32_t last = t + ((nblocks - 1 - t) / EV_T) * EV_T;
    acc = ld29_(part + last);
    if (last != (uint32_t)t) {
      const Fr29 big = tw29(pw[16 + LE]);
      for (int64_t k = (int64_t)last - EV_T; k >= t; k -= EV_T) acc = Fr29::add_csub(Fr29::mul(acc, big), ld29_(part + k));
    }
  }
  acc = eval_tree(acc, sh, pw + 8 + LE, t, levels);
  if (t == 0) stf(out + blockIdx.x, acc.to_fr());
}

// out[i] = sum_k s_k * P_k[i]  (+ constant at i == 0); the scalars are in twiddle form
struct LinTermD {
  const Fr* p;
  uint64_t len;
  Tw st;
};
struct LinCombArgsD {
  LinTermD t[24];
  int count;
  uint64_t len;
  Fr constant;
  Fr* out;
};
__global__ void __launch_bounds__(256) lincomb_kernel(LinCombArgsD a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.len) return;
  Fr29 acc = (i == 0) ? Fr29::from_fr(a.constant) : Fr29::zero();
  for (int k = 0; k < a.count; ++k)
    if (i < a.t[k].len) acc = Fr29::add_csub(acc, Fr29::mul(ld29_(a.t[k].p + i), tw29(a.t[k].st)));
  stf(a.out + i, acc.to_fr());
}

// ruffini: q_i = z^-(i+1) * sum_{j > i} c_j z^j
//   step 1: d_j = c_j z^j ; step 2: suffix sums ; step 3: q_i = S_{i+1} * zinv^(i+1)
// dst[i] = src[i + src_off] * x^(i + exp_off): a wave covers 64 * E consecutive elements, lane l
// takes l, l + 64, ... (coalesced) and steps its power by x^64.
// Round 6 (second session): the lane's first power x^(first + exp_off) is a product over the set bits of the exponent of
// the squarings x^(2^b), which the HOST computes once per call (a few microseconds of 64-bit arithmetic while the device
// runs the previous kernel) and passes in the kernel arguments — until then every lane ran its own square-and-multiply
// chain (~26 dependent products before its first element, 64 with the 32 of its 16 elements), which made the kernel pure
// latency below 2^18 coefficients: 40-44 us per call at 2^12 ... 2^16 gates and on every rank of a sharded proof, four calls
// per proof.  Now 6 products for the lane bits + one per set bit above them, and E follows the size (1 / 4 / 16 elements per
// lane) so that small arrays spread over the chip instead of serialising 16 elements per lane.
static constexpr int MP_BITS = 30;
struct MulPowArgs {
  Tw pw[MP_BITS];   // x^(2^b) in twiddle form, b < nbits
  Tw one_t;
  Fr addend;        // added to every source element before the multiplication (the suffix-sum carry of the ranks above)
  uint32_t nbits;   // bits of the largest exponent of the call
  uint64_t zero_at; // dst[zero_at] = 0 (ruffini's dropped slot; ~0: none) — saves a launch of its own
};
template <int E>
__global__ void __launch_bounds__(256) mul_powers_kernel(const Fr* __restrict__ src, Fr* __restrict__ dst, uint64_t n,
                                                         MulPowArgs a, uint64_t src_off, uint64_t exp_off) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g == 0 && a.zero_at != ~0ull) stf(dst + a.zero_at, Fr::zero());
  const uint64_t wave = g >> 6, lane = g & 63;
  const uint64_t first = wave * (64 * E) + lane;
  if (first >= n) return;
  const Fr29 add = Fr29::from_fr(a.addend);
  const uint64_t e = first + exp_off;
  Fr29 p = tw29(a.one_t);
  bool started = false;
#pragma unroll 1
  for (uint32_t b = 0; b < a.nbits; ++b) {
    if ((e >> b) & 1) {
      const Fr29 w = tw29(a.pw[b]);
      p = started ? Fr29::mul(p, w) : w;
      started = true;
    }
  }
  const Fr29 step = tw29(a.pw[6]);   // x^64
#pragma unroll 1
  for (int k = 0; k < E; ++k) {
    const uint64_t i = first + 64ull * k;
    if (i >= n) break;
    stf(dst + i, Fr29::mul(Fr29::add_csub(ld29_(src + i + src_off), add), p).to_fr());
    if (E > 1) p = Fr29::mul(p, step);
  }
}

// ---------------------------------------------------------------------------
// multi-GPU: the quotient coset split into residue classes (prover.hip, SURVEY §8e ii)
// ---------------------------------------------------------------------------
// dst[i] = src[i] + c * src[n + i] for i < extra, else src[i]: a polynomial of n + extra coefficients
// reduced mod (X^n - c), which is what it looks like on a size-n coset whose points satisfy x^n = c.
__global__ void fold_kernel(const Fr* __restrict__ src, Fr* __restrict__ dst, uint64_t n, uint32_t extra, Fr c) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr v = ldf(src + i);
  if (i < extra) v = v + c * ldf(src + n + i);
  stf(dst + i, v);
}

// send[(p * cpr + k) * stride + i] = F_k[p * per + i] (zero beyond n) for i < per, then F_k[0..8)
__global__ void shard_pack_kernel(ShardPackArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = blockIdx.y, peer = blockIdx.z;
  if (i >= a.stride) return;
  const Fr* F = a.F[k];
  Fr v = Fr::zero();
  if (i < a.per) {
    const uint64_t idx = (uint64_t)peer * a.per + i;
    if (idx < a.n) v = ldf(F + idx);
  } else {
    v = ldf(F + (i - a.per));
  }
  stf(a.send + ((uint64_t)peer * a.cpr + k) * a.stride + i, v);
}

// Per coefficient index i0 of this rank's range: the Q-point inverse DFT across the classes,
//   c_{i0 + n i1} = g^(-n i1) / Q * sum_j w_Q^(-j i1) F_j[i0]      (coef[i1][j] holds the factor),
// written to parts[i1][i0] (t_low, t_mid, t_high, t_fourth).  The last 8 lanes do the same for the
// lowest coefficients every rank received (de-aliasing / the coefficients of t beyond 4n).
__global__ void __launch_bounds__(128) shard_combine_kernel(ShardCombineArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.cnt + 8) return;
  const bool extra = i >= a.cnt;
  const uint64_t off = extra ? a.per + (i - a.cnt) : i;            // position inside a message
  Fr F[8];
  for (uint32_t j = 0; j < a.Q; ++j)
    F[j] = ldf(a.recv + ((uint64_t)(j % a.W) * a.cpr + j / a.W) * a.stride + off);
  auto dft = [&](uint32_t i1) {
    Fr acc = Fr::zero();
    for (uint32_t j = 0; j < a.Q; ++j) acc = acc + a.coef[i1][j] * F[j];
    return acc;
  };
  if (!extra) {
    const uint64_t idx = a.lo + i;
    for (uint32_t i1 = 0; i1 < 4; ++i1) {
      if (i1 == 0 && a.Q == 4 && idx < 7) continue;               // aliased: restored by the extra lanes
      stf(a.parts[i1] + idx, dft(i1));
    }
  } else {
    const uint64_t k = i - a.cnt;
    if (k >= 7) return;
    Fr top;                                                       // t[4n + k]
    if (a.Q == 4) {
      top = (dft(0) - a.low[k]) * a.g4n_inv;                      // A[k] = t[k] + g^4n t[4n + k]
      if (k >= a.lo && k < a.hi) stf(a.parts[0] + k, a.low[k]);
    } else {
      top = dft(4);
    }
    if (a.n + k >= a.lo && a.n + k < a.hi) stf(a.parts[3] + a.n + k, top);
  }
}

// blinding of the split quotient (prover.rs:547-574) restricted to the indices a rank owns
__global__ void shard_split_fix_kernel(ShardSplitFix a) {
  if (threadIdx.x != 0) return;
  auto own = [&](uint64_t idx) { return idx >= a.lo && idx < a.hi; };
  if (own(a.n)) { stf(a.parts[0] + a.n, a.b[0]); stf(a.parts[1] + a.n, a.b[1]); stf(a.parts[2] + a.n, a.b[2]); }
  if (own(0)) {
    stf(a.parts[1], ldf(a.parts[1]) - a.b[0]);
    stf(a.parts[2], ldf(a.parts[2]) - a.b[1]);
    stf(a.parts[3], ldf(a.parts[3]) - a.b[2]);
  }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
static inline dim3 grid1(uint64_t n, int t) { return dim3((uint32_t)((n + t - 1) / t)); }

int poly_fill_zero(Ctx* c, Fr* p, uint64_t n) {
  if (!n) return PLONK_OK;
  hipLaunchKernelGGL(fill_zero_kernel, grid1(n, 256), dim3(256), 0, c->stream, p, n);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_blind(Ctx* c, Fr* coeffs, uint64_t n, const BlindArgs& a) {
  hipLaunchKernelGGL(blind_kernel, dim3(1), dim3(64), 0, c->stream, coeffs, n, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_split_t(Ctx* c, Fr* t, uint64_t n, uint64_t np, Fr* out, const SplitArgs& a) {
  hipLaunchKernelGGL(split_t_kernel, dim3((uint32_t)((np + 255) / 256), 3), dim3(256), 0, c->stream, t, n, np, out, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_trimmed_len(Ctx* c, const Fr* p, uint64_t n, unsigned long long* out_dev) {
  HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(trimmed_len_kernel, grid1(n, 256), dim3(256), 0, c->stream, p, n, out_dev);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_sigma_evals(Ctx* c, const uint32_t* map_dev, const Fr* roots, Fr* out, uint64_t n, uint64_t stride, const SigmaArgs& a) {
  dim3 grid = grid1(n, 256);
  grid.y = 4;
  hipLaunchKernelGGL(sigma_evals_kernel, grid, dim3(256), 0, c->stream, map_dev, roots, out, n, stride, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_gather_wires(Ctx* c, const uint32_t* idx_dev, const Fr* values, Fr* wires, uint64_t constraints, uint64_t n) {
  dim3 grid = grid1(n, 256);
  grid.y = 4;
  hipLaunchKernelGGL(gather_wires_kernel, grid, dim3(256), 0, c->stream, idx_dev, values, wires, constraints, n);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_scatter_pi(Ctx* c, Fr* dense, const uint64_t* idx, const Fr* val, uint64_t count) {
  if (!count) return PLONK_OK;
  hipLaunchKernelGGL(scatter_pi_kernel, grid1(count, 256), dim3(256), 0, c->stream, dense, idx, val, count);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_batch_inverse(Ctx* c, Fr* v, uint64_t n, bool twiddle_form) {
  static const BatchInvArgs a = [] {
    BatchInvArgs r;
    r.one_t = tw_of(Fr::one());
    r.one_r = tw_plain(Fr::one());
    // R''^2 / R = (R''/R) in twiddle form; R''/R = 2^5
    r.conv = tw_of(Fr::from_u64(32));
    return r;
  }();
  // geometry: lanes x elements per workgroup.  Up to 2^17 elements (every proof below 2^18 gates, and a rank's range of a
  // sharded grand product) one wave x 4 elements: the array spreads over 4x as many CUs and the scans need no workgroup
  // barrier (2^16 gates: -27 us on the critical path of round 2, 2^12: -25 us); above, 256 x 16 does 16x fewer inversions
  // (2^20: 32.81 against 33.06 ms).  Same results either way; PLONK_BI_CFG=0..3 forces one geometry (A/B runs,
  // profiles/r04/SUMMARY.md section 8).
  const int cfg_env = c->cfg.bi_cfg;
  const int cfg = cfg_env >= 0 && cfg_env <= 3 ? cfg_env : (n <= (1ull << 17) ? 1 : 0);
#define BI_LAUNCH(T, E)                                                                                                              \
  do {                                                                                                                               \
    if (twiddle_form) hipLaunchKernelGGL((batch_inverse_kernel<true, T, E>), grid1(n, T * E), dim3(T), 0, c->stream, v, n, a);         \
    else hipLaunchKernelGGL((batch_inverse_kernel<false, T, E>), grid1(n, T * E), dim3(T), 0, c->stream, v, n, a);                     \
  } while (0)
  if (cfg == 1) BI_LAUNCH(64, 4);
  else if (cfg == 2) BI_LAUNCH(64, 16);
  else if (cfg == 3) BI_LAUNCH(128, 8);
  else BI_LAUNCH(256, 16);
#undef BI_LAUNCH
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_perm_terms(Ctx* c, const PermArgs& a) {
  PermConst k;
  k.conv = tw_of(Fr::from_u64(32));
  k.gamma_t = tw_of(a.gamma);
  for (int j = 0; j < 4; ++j) k.bk_t[j] = tw_of(a.beta * a.ks[j]);
  k.b32_t = tw_of(a.beta * Fr::from_u64(32));
  k.one_t = tw_of(Fr::one());
  PermArgs r = a;
  if (r.count == 0) { r.first = 0; r.count = a.n; }
  if (r.first + r.count > a.n) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  hipLaunchKernelGGL(perm_terms_kernel, grid1(r.count, 128), dim3(128), 0, c->stream, r, k);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_mul_arrays(Ctx* c, Fr* a, const Fr* b, uint64_t n, int* zero_flag) {
  hipLaunchKernelGGL(mul_arrays_kernel, grid1(n, 256), dim3(256), 0, c->stream, a, b, n, zero_flag);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_quotient(Ctx* c, const QuotientArgs& q) {
  prof_begin(c, 3);
  WidgetConst wc{};
  if (q.has[QS_RANGE] | q.has[QS_LOGIC] | q.has[QS_FIXED] | q.has[QS_VAR]) {
    wc.conv = tw_of(Fr::from_u64(32));
    wc.one_r = tw_plain(Fr::one());
    wc.c1 = tw_of(Fr::one()); wc.c3 = tw_of(Fr::from_u64(3)); wc.c9 = tw_of(Fr::from_u64(9));
    wc.c18 = tw_of(Fr::from_u64(18)); wc.c81 = tw_of(Fr::from_u64(81)); wc.c83 = tw_of(Fr::from_u64(83));
    wc.ed = tw_of(q.edwards_d);
    Fr k = q.range_ch.sqr(), w = q.range_ch;
    for (int j = 0; j < 4; ++j) { wc.rg[j] = tw_of(w); w = w * k; }
    k = q.logic_ch.sqr();
    const Fr lk[5] = {k.sqr() * k, Fr::one(), k, k.sqr(), k.sqr().sqr()};   // k^3, 1, k, k^2, k^4
    for (int j = 0; j < 5; ++j) wc.lg[j] = tw_of(lk[j] * q.logic_ch);
    k = q.fixed_ch.sqr(); w = q.fixed_ch;
    for (int j = 0; j < 4; ++j) { wc.fx[j] = tw_of(w); w = w * k; }
    k = q.var_ch.sqr(); w = q.var_ch;
    for (int j = 0; j < 3; ++j) { wc.vr[j] = tw_of(w); w = w * k; }
    hipLaunchKernelGGL(quotient_kernel<true>, grid1(q.n8, 128), dim3(128), 0, c->stream, q, wc);
  } else {
    hipLaunchKernelGGL(quotient_kernel<false>, grid1(q.n8, 128), dim3(128), 0, c->stream, q, wc);
  }
  prof_end(c, 3);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_from_mont(Ctx* c, const Fr* src, Fr* dst, uint64_t n) {
  if (!n) return PLONK_OK;
  hipLaunchKernelGGL(from_mont_kernel, grid1(n, 256), dim3(256), 0, c->stream, src, dst, n);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_scale_array(Ctx* c, Fr* v, uint64_t n, const Fr& s) {
  hipLaunchKernelGGL(scale_array_kernel, grid1(n, 256), dim3(256), 0, c->stream, v, n, s);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
// host helper: challenge constant c (Fr, R form) * 2^shift  ->  limbs of c * 2^shift * R'' (fr29.cuh)
void quotient_const(const Fr& c, int shift, uint32_t out[9]) {
  Fr v = c;
  for (int k = 0; k < shift; ++k) v = v.dbl();
  const Fr29 r = Fr29::twiddle_from_fr(v);
  for (int k = 0; k < 9; ++k) out[k] = r.l[k];
}
void quotient_data(const Fr& c, uint32_t out[9]) {   // plain re-slicing (stays in R form)
  const Fr29 r = Fr29::from_fr(c);
  for (int k = 0; k < 9; ++k) out[k] = r.l[k];
}
int poly_dealias(Ctx* c, Fr* t, uint64_t nq, const Fr low[7], const Fr& g_inv) {
  DealiasArgs a;
  for (int k = 0; k < 7; ++k) a.low[k] = low[k];
  a.g_inv = g_inv;
  hipLaunchKernelGGL(dealias_kernel, dim3(1), dim3(64), 0, c->stream, t, nq, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_l1(Ctx* c, const Fr* linear, Fr* l1, uint64_t n8, const L1Args& a) {
  hipLaunchKernelGGL(l1_prepare_kernel, grid1(n8, 256), dim3(256), 0, c->stream, linear, l1, n8);
  int rc = poly_batch_inverse(c, l1, n8);
  if (rc) return rc;
  hipLaunchKernelGGL(l1_finish_kernel, grid1(n8, 256), dim3(256), 0, c->stream, l1, n8, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_eval(Ctx* c, EvalArgs& a, int count, uint64_t max_len, Fr* out_dev) {
  if (count > 16 || max_len == 0) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  // coefficients per lane by size: 4 up to 2^17 (the chain of dependent products is what a small evaluation costs), else 16
  const int le = max_len <= (1ull << 17) ? 2 : 4;
  const uint64_t per_wg = (uint64_t)EV_T << le;
  const uint32_t nb = (uint32_t)((max_len + per_wg - 1) / per_wg);
  if (nb > a.max_blocks || (uint64_t)nb > 256ull * 64) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);   // (x^(2^(16 + LE)) is the last power: the lanes' own Horner takes the rest)
  EvalArgsD d;
  Fr pts[EV_POINTS];
  int npts = 0;
  for (int k = 0; k < count; ++k) {
    d.items[k].poly = a.items[k].poly;
    d.items[k].len = a.items[k].len;
    int w = 0;
    while (w < npts && !(pts[w] == a.items[k].x)) ++w;
    if (w == npts) {
      if (npts == EV_POINTS) return (plonk::set_last_error("invalid argument", "poly_eval: more than 2 distinct points", __FILE__, __LINE__), PLONK_ERR_ARG);
      pts[npts++] = a.items[k].x;
      Fr29 p = Fr29::twiddle_from_fr(a.items[k].x);
      for (int b = 0; b < EV_PW; ++b) {
        for (int i = 0; i < 9; ++i) d.pw[w][b].w[i] = p.l[i];
        p = Fr29::mul(p, p);
      }
    }
    d.items[k].point = (uint32_t)w;
  }
  d.partial = a.partial;
  d.max_blocks = a.max_blocks;
  int levels = 0;
  while (levels < 8 && (1u << levels) < nb) ++levels;
  if (le == 2) {
    hipLaunchKernelGGL(eval_kernel<2>, dim3(nb, count), dim3(EV_T), 0, c->stream, d);
    hipLaunchKernelGGL(eval_final_kernel<2>, dim3(count), dim3(EV_T), 0, c->stream, d, nb, levels, out_dev);
  } else {
    hipLaunchKernelGGL(eval_kernel<4>, dim3(nb, count), dim3(EV_T), 0, c->stream, d);
    hipLaunchKernelGGL(eval_final_kernel<4>, dim3(count), dim3(EV_T), 0, c->stream, d, nb, levels, out_dev);
  }
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_lincomb(Ctx* c, const LinCombArgs& a) {
  LinCombArgsD d;
  for (int k = 0; k < a.count; ++k) {
    d.t[k].p = a.t[k].p;
    d.t[k].len = a.t[k].len;
    d.t[k].st = tw_of(a.t[k].s);
  }
  d.count = a.count;
  d.len = a.len;
  d.constant = a.constant;
  d.out = a.out;
  hipLaunchKernelGGL(lincomb_kernel, grid1(a.len, 256), dim3(256), 0, c->stream, d);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
static void launch_mul_powers(Ctx* c, const Fr* src, Fr* dst, uint64_t n, const Fr& x, uint64_t src_off, uint64_t exp_off,
                              const Fr& addend = Fr::zero(), uint64_t zero_at = ~0ull) {
  if (!n) {
    if (zero_at != ~0ull) (void)poly_fill_zero(c, dst + zero_at, 1);
    return;
  }
  MulPowArgs a;
  const uint64_t emax = n - 1 + exp_off;
  uint32_t nbits = 7;                                   // pw[6] = x^64 is the lanes' step
  while (nbits < MP_BITS && (emax >> nbits)) ++nbits;   // (exponents stay below 2^30: n <= 2^25 + a range offset)
  Fr29 p = Fr29::twiddle_from_fr(x);
  for (uint32_t b = 0; b < nbits; ++b) {
    for (int i = 0; i < 9; ++i) a.pw[b].w[i] = p.l[i];
    p = Fr29::mul(p, p);                                // twiddle form is closed under the product: (x R'')^2 / R'' = x^2 R''
  }
  a.one_t = tw_of(Fr::one());
  a.addend = addend;
  a.nbits = nbits;
  a.zero_at = zero_at;
  // elements per lane by size: the kernel is latency below ~2^18 coefficients (one product chain per lane), throughput above
  if (n <= (1ull << 17)) hipLaunchKernelGGL(mul_powers_kernel<1>, grid1(n, 256), dim3(256), 0, c->stream, src, dst, n, a, src_off, exp_off);
  else if (n <= (1ull << 19)) hipLaunchKernelGGL(mul_powers_kernel<4>, grid1((n + 3) / 4 + 64, 256), dim3(256), 0, c->stream, src, dst, n, a, src_off, exp_off);
  else hipLaunchKernelGGL(mul_powers_kernel<16>, grid1((n + 15) / 16 + 64, 256), dim3(256), 0, c->stream, src, dst, n, a, src_off, exp_off);
}
// quotient of src[0..len) by (X - z) -> dst[0..len-1); dst[len-1] = 0.  scratch: len Fr + totals.
int poly_ruffini(Ctx* c, const Fr* src, Fr* dst, uint64_t len, const Fr& z, const Fr& zinv, Fr* scratch, Fr* totals) {
  launch_mul_powers(c, src, scratch, len, z, 0, 0);
  int rc = scan_suffix_sum(c, scratch, len, totals);
  if (rc) return rc;
  // q_i = S_{i+1} * zinv^(i+1), i < len - 1
  launch_mul_powers(c, scratch, dst, len - 1, zinv, 1, 1, Fr::zero(), len - 1);   // (+ dst[len - 1] = 0)
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

int poly_fold(Ctx* c, const Fr* src, Fr* dst, uint64_t n, uint32_t extra, const Fr& cn) {
  hipLaunchKernelGGL(fold_kernel, grid1(n, 256), dim3(256), 0, c->stream, src, dst, n, extra, cn);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_shard_pack(Ctx* c, const ShardPackArgs& a, uint32_t world) {
  hipLaunchKernelGGL(shard_pack_kernel, dim3((uint32_t)((a.stride + 255) / 256), a.cpr, world), dim3(256), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_shard_combine(Ctx* c, const ShardCombineArgs& a) {
  hipLaunchKernelGGL(shard_combine_kernel, grid1(a.cnt + 8, 128), dim3(128), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int poly_shard_split_fix(Ctx* c, const ShardSplitFix& a) {
  hipLaunchKernelGGL(shard_split_fix_kernel, dim3(1), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
// Range-sharded ruffini, part 1: scratch[i] = src[i] * z^(lo + i), suffix sums in place, scratch[len] = 0;
// scratch[0] is then this range's contribution to sum_j c_j z^j.
int poly_ruffini_local(Ctx* c, const Fr* src, uint64_t lo, uint64_t len, const Fr& z, Fr* scratch, Fr* totals) {
  int rc = poly_fill_zero(c, scratch + len, 1);
  if (rc || !len) return rc;
  launch_mul_powers(c, src, scratch, len, z, 0, lo);
  return scan_suffix_sum(c, scratch, len, totals);
}
// part 2: dst[lo + i] = (scratch[i + 1] + carry) * zinv^(lo + i + 1) for i < len, where carry = the
// contributions of all higher ranges; index `last` (the dropped remainder slot) is set to zero.
int poly_ruffini_finish(Ctx* c, const Fr* scratch, Fr* dst, uint64_t lo, uint64_t len, const Fr& zinv, const Fr& carry, uint64_t last) {
  launch_mul_powers(c, scratch, dst + lo, len, zinv, 1, lo + 1, carry);
  if (last >= lo && last < lo + len) return poly_fill_zero(c, dst + last, 1);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

}  // namespace plonk
