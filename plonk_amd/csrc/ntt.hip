// Radix-2 NTT / iNTT / coset variants over BLS12-381 Fr for gfx950 (MI355X).
//
// Replaces the bodies of EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
// (reference src/fft/domain.rs:166-232) i.e. best_fft (:383-422) + bitreverse
// (:434) + the n^-1 scale (:195) + distribute_powers (:198-204).  Results are the
// same field elements bit for bit (an NTT is a unique linear map); only the
// schedule differs:
//
//   N = R1 * R2 * R3 (each <= 512).  Pass A does R1-point transforms down the
//   most significant index digit (stride N/R1), multiplies by w_N^(k1 * col) and
//   stores TRANSPOSED; pass B (3-pass only) and pass C run the remaining digits
//   in place.  Natural order in, natural order out, no bit-reversal pass
//   (Stockham-style autosort through the transposed store).  Every pass moves a
//   [R rows][C cols] tile of 256 * E elements per 256-thread workgroup: global
//   loads/stores are runs of C (>= 4) consecutive 32-byte elements, the
//   transposed store writes runs of R1 consecutive elements.
//
//   Inside a tile each thread keeps E elements and performs radix-E decimation-in-frequency
//   rounds in registers, in the reduced-radix lazy form of fr29.cuh (9 x 29-bit limbs: a
//   butterfly is ~290 VALU instructions, 162 of them v_mad_u64_u32, instead of ~890 with
//   32-bit limbs and carry chains).  Rounds exchange data through LDS kept limb-planar
//   (9 planes of u32) so DS traffic is plain 4-byte accesses.  E = 4 (1024-element tiles, 36 KiB,
//   four waves per SIMD) and E = 8 (2048-element tiles, 72 KiB, two waves) are both compiled:
//   NTT_ELOG_DEFAULT below.  Small twiddles (w_512^e) come from an L1-resident table; the inter-pass
//   twiddle w_N^e is TWLO[e & 8191] * TWHI[e >> 13].  Tables hold w * 2^261 (fr29.cuh) so
//   data stays in the reference's Montgomery domain; elements are converted to canonical
//   32-bit limbs only when a pass stores to HBM.
//
// HBM traffic: 64*N bytes per pass (32 read + 32 written), i.e. 128*N / 192*N
// for the 2- / 3-pass plans against the algorithmic 64*N.  The kernel is bound by
// 32-bit integer multiply issue (one Fr product ~ 128 v_mad_u64_u32), not HBM.
//
// tests/ntt_model.py mirrors the index arithmetic below (same names) and is
// checked against the oracle on CPU.
#include "plonk_internal.hpp"
#include "fr29.cuh"
#include <type_traits>
#include <utility>

namespace plonk {

static constexpr int NTT_THREADS = 256;
static constexpr int NTT_THREAD_BITS = 8;
// Elements per lane: 2^ELOG, tile = 2^(8 + ELOG) elements per 256-thread workgroup.  ELOG = 3 (8 elements, radix-8 register
// rounds, 2048-element tiles, 72 KiB of LDS, ~234 VGPRs: two waves per SIMD) is the round 1-3 kernel; ELOG = 2 (round 4:
// 4 elements, radix-4 rounds, 1024-element tiles, 36 KiB, <= 128 VGPRs: four waves per SIMD) trades one more LDS exchange
// per pass for twice the waves to cover the strided tile loads and twiddle gathers: the same instructions per element
// (+0.6 %), transforms alone 6-10 % faster from 2^20 to 2^24 and 44 % at 2^16 (twice the workgroups for a chip that a
// 2^16-point pass does not fill) — profiles/r04/log_r4v.txt.  It is the default; the one place where it LOSES is side-stream
// work issued under a busy MSM pipeline (prover.hip SideScope): at <= 128 VGPRs the passes co-reside with the sort and
// reduction kernels of the main stream and take issue slots from the critical path (2^20 gates: prove() 32.8 -> 33.4 ms,
// 2^22: 120.3 -> 122.3), so that scope asks for ELOG = 3 through Ctx::ntt_elog_hint.  PLONK_NTT_ELOG=2|3 forces either.
static constexpr int NTT_ELOG_DEFAULT = 2;
static constexpr int TWLO_BITS = 13;
static constexpr int GLO_BITS = 10;
static constexpr int NTT_DIRECT_MAX_LOG = 25;

struct NttPass {
  const Fr* src;
  Fr* dst;
  uint32_t logN;
  // element address = row * rs + (cg >> hshift) * hs + (cg & ((1<<hshift)-1)) * ls
  uint64_t in_rs, in_hs;
  uint32_t in_hshift;
  uint64_t out_rs, out_hs, out_ls;
  uint32_t out_hshift;
  // inter-pass twiddle: value *= w_N^(k * ((cg >> tw_shr) << tw_shr))
  int tw_mode;
  uint32_t tw_shr;
  const Fr29Slot* tw_lo;
  const Fr29Slot* tw_hi;
  // when set: the twiddle is read whole from tw_direct[e >> tw_shr] (pass A: w_N^e for every e < N; pass B: the N / R1
  // multiples of R1 that its exponents can take) instead of the two-level product — one Fr product less per element
  const Fr29Slot* tw_direct;
  // first pass: zero beyond in_len, optional coset scale g^i
  uint64_t in_len;
  int pre_coset;
  // last pass: 0 none, 1 multiply by `scale`, 2 multiply by GLO[o & 1023] * GHI[o >> 10]
  int post_mode;
  Fr29 scale;
  const Fr29Slot* g_lo;
  const Fr29Slot* g_hi;
  const Fr29Slot* w512;   // w_512^e, e < 256 (direction specific), L1-resident
};

__device__ __forceinline__ Fr ld_fr(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void st_fr(Fr* p, const Fr& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// LDS planes: data[l * 2048 + idx]
__device__ __forceinline__ Fr29 lds_get(const uint32_t* base, int stride, int idx) {
  Fr29 r;
#pragma unroll
  for (int l = 0; l < Fr29::N; ++l) r.l[l] = base[l * stride + idx];
  return r;
}
__device__ __forceinline__ void lds_put(uint32_t* base, int stride, int idx, const Fr29& v) {
#pragma unroll
  for (int l = 0; l < Fr29::N; ++l) base[l * stride + idx] = v.l[l];
}
__device__ __forceinline__ Fr29 ld_tw(const Fr29Slot* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  const uint32_t c = p->w[8];
  Fr29 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c;
  return r;
}
__device__ __forceinline__ void st_tw(Fr29Slot* p, const Fr29& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  q[2] = make_uint4(v.l[8], 0u, 0u, 0u);
}

// tests/ntt_model.py: elem_index
template <int POS, int RB>
__device__ __forceinline__ int elem_index(int t, int e) {
  const int j = e & ((1 << RB) - 1);
  const int ge = e >> RB;
  const int rest = (ge << NTT_THREAD_BITS) | t;
  const int lo = rest & ((1 << POS) - 1);
  const int hi = rest >> POS;
  return (hi << (POS + RB)) | (j << POS) | lo;
}

// One DIF stage on local bit LB of a register round (global row bit LO + LB).
template <int RLOG, int LO, int LB, int E>
__device__ __forceinline__ void dif_stage(Fr29 (&v)[E], int rlow_thread, const Fr29Slot* __restrict__ wtab) {
  constexpr int bitpos = LO + LB;
#pragma unroll
  for (int x = 0; x < (1 << LB); ++x) {
    // The twiddle is w_512^(rlow * 2^(8-bitpos)); in the round on the lowest row bits (LO == 0)
    // rlow == x is a compile-time constant after unrolling, and x == 0 means w = 1: those
    // butterflies (3 more per lane and pass besides the last stage) need no multiplication.
    const bool trivial = bitpos == 0 || (LO == 0 && x == 0);
    Fr29 w;
    if (!trivial) {
      const int rlow = rlow_thread | (x << LO);          // row & (2^bitpos - 1)
      w = ld_tw(wtab + (rlow << (8 - bitpos)));
    }
#pragma unroll
    for (int y = 0; y < (E >> (LB + 1)); ++y) {         // bits above LB (incl. extra groups)
      const int e = (y << (LB + 1)) | x;
      const int e2 = e | (1 << LB);
      const Fr29 a = v[e], b = v[e2];
      v[e] = Fr29::add_csub(a, b);
      if (!trivial) v[e2] = Fr29::mul(Fr29::sub_lazy(a, b), w);
      else v[e2] = Fr29::sub_reduce(a, b);
    }
  }
}

// One register round: RB DIF stages on the row bits [LO, LO+RB), top bit first.
template <int RLOG, int LO, int RB, int ELOG>
__device__ __forceinline__ void dif_round(Fr29 (&v)[1 << ELOG], int t, const Fr29Slot* __restrict__ wtab) {
  constexpr int CLOG = NTT_THREAD_BITS + ELOG - RLOG;
  static_assert(RB <= ELOG && CLOG + LO <= NTT_THREAD_BITS, "the row bits below a round come from the thread index");
  const int rlow_thread = (LO > 0) ? ((t >> CLOG) & ((1 << LO) - 1)) : 0;
  if constexpr (RB >= 3) dif_stage<RLOG, LO, 2, (1 << ELOG)>(v, rlow_thread, wtab);
  if constexpr (RB >= 2) dif_stage<RLOG, LO, 1, (1 << ELOG)>(v, rlow_thread, wtab);
  dif_stage<RLOG, LO, 0, (1 << ELOG)>(v, rlow_thread, wtab);
}

template <int RLOG, int R_IDX, int ELOG>
struct RoundGeom {
  static constexpr int HI = RLOG - ELOG * R_IDX;
  static constexpr int LO = (HI - ELOG > 0) ? HI - ELOG : 0;
  static constexpr int RB = HI - LO;
  static constexpr int POS = (NTT_THREAD_BITS + ELOG - RLOG) + LO;
};

template <int RLOG, int R_IDX, int NR, int ELOG>
struct Rounds {
  // rounds R_IDX .. NR-1; on entry v holds round R_IDX's elements
  __device__ static __forceinline__ void run(Fr29 (&v)[1 << ELOG], int t, uint32_t* data, const Fr29Slot* __restrict__ wtab) {
    constexpr int TILE_LOG = NTT_THREAD_BITS + ELOG;
    using G = RoundGeom<RLOG, R_IDX, ELOG>;
    dif_round<RLOG, G::LO, G::RB, ELOG>(v, t, wtab);
    if constexpr (R_IDX + 1 < NR) {
      using Gn = RoundGeom<RLOG, R_IDX + 1, ELOG>;
      // a lane stores to the very indices it loaded this round's operands from: no barrier needed before the stores
#pragma unroll
      for (int e = 0; e < (1 << ELOG); ++e) lds_put(data, 1 << TILE_LOG, elem_index<G::POS, G::RB>(t, e), v[e]);
      __syncthreads();
#pragma unroll
      for (int e = 0; e < (1 << ELOG); ++e) v[e] = lds_get(data, 1 << TILE_LOG, elem_index<Gn::POS, Gn::RB>(t, e));
      Rounds<RLOG, R_IDX + 1, NR, ELOG>::run(v, t, data, wtab);
    }
  }
};

__device__ __forceinline__ Fr two_level_fr(const Fr* lo, const Fr* hi, uint64_t e, int lobits, bool use_hi) {
  Fr w = ld_fr(lo + (e & ((1ull << lobits) - 1)));
  if (use_hi) w = w * ld_fr(hi + (e >> lobits));
  return w;
}
__device__ __forceinline__ Fr29 two_level(const Fr29Slot* lo, const Fr29Slot* hi, uint64_t e, int lobits, bool use_hi) {
  Fr29 w = ld_tw(lo + (e & ((1ull << lobits) - 1)));
  if (use_hi) w = Fr29::mul(w, ld_tw(hi + (e >> lobits)));
  return w;
}
// compile-time loop over the E elements of a lane: f(std::integral_constant<int, e>) for e = 0 .. E - 1
template <typename F, int... Is>
__device__ __forceinline__ void for_elems_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int E, typename F>
__device__ __forceinline__ void for_elems(F&& f) {
  for_elems_impl(static_cast<F&&>(f), std::make_integer_sequence<int, E>{});
}

template <int RLOG, bool TRANSPOSE, int ELOG>
__global__ void __launch_bounds__(NTT_THREADS, ELOG == 2 ? 4 : 2) ntt_pass_kernel(NttPass p) {
  constexpr int E = 1 << ELOG;
  constexpr int TILE_LOG = NTT_THREAD_BITS + ELOG;
  constexpr int CLOG = TILE_LOG - RLOG;
  constexpr int C = 1 << CLOG;
  constexpr int NR = (RLOG + ELOG - 1) / ELOG;
  static_assert(CLOG >= 1 && RLOG <= 9, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* data = smem;                       // 9 * 2^TILE_LOG u32
  const Fr29Slot* __restrict__ wtab = p.w512;
  const int t = threadIdx.x;
  const uint64_t cg0 = (uint64_t)blockIdx.x * C;

  // ---- load round 0 operands straight from HBM.  All eight loads of a lane are issued back to back in one branch-free
  // block (index clamped to 0 beyond in_len, the value masked to zero afterwards): with a branch per element the
  // compiler waited for each load before issuing the next — eight exposed HBM latencies per wave at two waves per SIMD
  // (r03e: the passes ran at 5.7-6.4 cycles per VALU instruction against 4.7 in msm_accumulate).
  // for_elems is a compile-time loop: `#pragma unroll` gives up on bodies of this size (a Montgomery product is ~600
  // instructions) and a rolled loop would index the arrays in scratch memory.
  using G0 = RoundGeom<RLOG, 0, ELOG>;
  Fr29 v[E];
  {
    Fr raw[E];
    uint64_t gis[E];
    for_elems<E>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      const int idx = elem_index<G0::POS, G0::RB>(t, e);
      const uint64_t row = idx >> CLOG;
      const uint64_t cg = cg0 + (idx & (C - 1));
      gis[e] = row * p.in_rs + (cg >> p.in_hshift) * p.in_hs + (cg & ((1ull << p.in_hshift) - 1));
      raw[e] = ld_fr(p.src + (gis[e] < p.in_len ? gis[e] : 0));   // in_len >= 1 (ntt_device)
    });
    for_elems<E>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      const uint32_t keep = gis[e] < p.in_len ? 0xffffffffu : 0u;
      const Fr29 x = Fr29::from_fr(raw[e]);
#pragma unroll
      for (int l = 0; l < Fr29::N; ++l) v[e].l[l] = x.l[l] & keep;
    });
    if (p.pre_coset) {   // coset scale g^i of the valid coefficients (n + 3 of 4n in the prover: whole rows are skipped)
      for_elems<E>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        if (gis[e] < p.in_len) v[e] = Fr29::mul(v[e], two_level(p.g_lo, p.g_hi, gis[e], GLO_BITS, true));
      });
    }
  }
  Rounds<RLOG, 0, NR, ELOG>::run(v, t, data, wtab);

  // ---- epilogue: inter-pass twiddle / scaling, then store.  The pass-wide modes are tested OUTSIDE the element loops so
  // that each loop is one basic block: the eight twiddle loads (HBM gathers with the direct table of pass A) go out
  // together and the multiplications follow.
  using GL = RoundGeom<RLOG, NR - 1, ELOG>;
  const uint64_t nmask = (1ull << p.logN) - 1;
  const bool use_hi = p.logN > TWLO_BITS;
  uint32_t ks[E];
  uint64_t cgs[E];
  for_elems<E>([&](auto ec) __attribute__((always_inline)) {
    constexpr int e = decltype(ec)::value;
    const int idx = elem_index<GL::POS, GL::RB>(t, e);
    const uint32_t prow = idx >> CLOG;
    ks[e] = __brev(prow) >> (32 - RLOG);
    cgs[e] = cg0 + (idx & (C - 1));
  });
  if (p.tw_mode) {
    if (p.tw_direct) {
      Fr29 tw[E];
      for_elems<E>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        const uint64_t twcol = (cgs[e] >> p.tw_shr) << p.tw_shr;
        tw[e] = ld_tw(p.tw_direct + ((((uint64_t)ks[e] * twcol) & nmask) >> p.tw_shr));   // tests/ntt_model.py: direct_index
      });
      for_elems<E>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        v[e] = Fr29::mul(v[e], tw[e]);
      });
    } else {
      for_elems<E>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        const uint64_t twcol = (cgs[e] >> p.tw_shr) << p.tw_shr;
        v[e] = Fr29::mul(v[e], two_level(p.tw_lo, p.tw_hi, ((uint64_t)ks[e] * twcol) & nmask, TWLO_BITS, use_hi));
      });
    }
  }
  if constexpr (!TRANSPOSE) {
    uint64_t os[E];
    for_elems<E>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      os[e] = (uint64_t)ks[e] * p.out_rs + (cgs[e] >> p.out_hshift) * p.out_hs + (cgs[e] & ((1ull << p.out_hshift) - 1)) * p.out_ls;
    });
    if (p.post_mode == 1) {
      for_elems<E>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        v[e] = Fr29::mul(v[e], p.scale);
      });
    } else if (p.post_mode == 2) {
      for_elems<E>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        v[e] = Fr29::mul(v[e], two_level(p.g_lo, p.g_hi, os[e], GLO_BITS, true));
      });
    }
    for_elems<E>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      st_fr(p.dst + os[e], v[e].to_fr());
    });
  } else {
    constexpr int R = 1 << RLOG;
    __syncthreads();   // everyone finished reading `data` for the last round
    for_elems<E>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      lds_put(data, 1 << TILE_LOG, (int)(cgs[e] - cg0) * R + (int)ks[e], v[e]);
    });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int u = i * NTT_THREADS + t;
      const int col = u >> RLOG;
      const uint64_t k = u & (R - 1);
      const uint64_t cg = cg0 + col;
      const Fr29 x = lds_get(data, 1 << TILE_LOG, u);
      const uint64_t o = k * p.out_rs + (cg >> p.out_hshift) * p.out_hs +
                         (cg & ((1ull << p.out_hshift) - 1)) * p.out_ls;
      st_fr(p.dst + o, x.to_fr());
    }
  }
}

// ---- small transforms (N <= 1024): one workgroup, DIT on a bit-reversed load,
// twiddles straight from the full w_N^e table (TWLO holds all of them for L <= 13).
struct NttSmall {
  const Fr* src;
  Fr* dst;
  uint32_t logN;
  uint64_t in_len;
  int pre_coset, post_mode;
  Fr scale;
  const Fr* tw_lo;
  const Fr* g_lo;
  const Fr* g_hi;
};

__global__ void __launch_bounds__(NTT_THREADS) ntt_small_kernel(NttSmall p) {
  __shared__ __attribute__((aligned(16))) Fr buf[1024];
  const uint32_t L = p.logN, N = 1u << L;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) {
    const uint32_t s = L ? (__brev(i) >> (32 - L)) : 0;
    Fr x = Fr::zero();
    if (s < p.in_len) {
      x = ld_fr(p.src + s);
      if (p.pre_coset) x = x * two_level_fr(p.g_lo, p.g_hi, s, GLO_BITS, true);
    }
    buf[i] = x;
  }
  __syncthreads();
  for (uint32_t s = 0; s < L; ++s) {
    const uint32_t m = 1u << s;
    for (uint32_t b = threadIdx.x; b < N / 2; b += blockDim.x) {
      const uint32_t j = b & (m - 1);
      const uint32_t lo = ((b >> s) << (s + 1)) | j;
      const uint32_t hi = lo + m;
      Fr tt = buf[hi];
      if (j) tt = tt * ld_fr(p.tw_lo + ((uint64_t)j << (L - 1 - s)));
      Fr a = buf[lo];
      buf[hi] = a - tt;
      buf[lo] = a + tt;
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) {
    Fr x = buf[i];
    if (p.post_mode) x = x * p.scale;                                                    // n^-1
    if (p.post_mode == 2) x = x * two_level_fr(p.g_lo, p.g_hi, i, GLO_BITS, true);       // g^-i
    st_fr(p.dst + i, x);
  }
}

// out[i] = first * base^(i * step_mul)   (table builder)
__global__ void pow_table_kernel(Fr* out, Fr base, Fr first, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  st_fr(out + i, first * base.pow_u64(i));
}

// same, stored as w * 2^261 in 29-bit limbs (fr29.cuh)
__global__ void pow_table29_kernel(Fr29Slot* out, Fr base, Fr first, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  st_tw(out + i, Fr29::twiddle_from_fr(first * base.pow_u64(i)));
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
void ntt_plan(uint32_t L, int r[3], int* npass) {   // tests/ntt_model.py: plan
  if (L <= 10) { *npass = 1; r[0] = (int)L; return; }
  if (L <= 18) { *npass = 2; r[0] = (int)(L + 1) / 2; r[1] = (int)L - r[0]; return; }
  *npass = 3;
  r[0] = (int)(L + 2) / 3;
  r[1] = ((int)L - r[0] + 1) / 2;
  r[2] = (int)L - r[0] - r[1];
}

static Fr host_omega(uint32_t L, bool inverse) {   // domain.rs:142-145,155
  Fr g = fr_root_of_unity();
  for (uint32_t i = L; i < 32; ++i) g = g.sqr();
  return inverse ? g.inv() : g;
}

static int build_pow_table(Ctx* c, Fr** out, const Fr& base, const Fr& first, uint32_t count) {
  HIP_TRY(hipMalloc((void**)out, sizeof(Fr) * (size_t)count));
  hipLaunchKernelGGL(pow_table_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, *out, base, first, count);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

static int build_pow_table29(Ctx* c, Fr29Slot** out, const Fr& base, const Fr& first, uint32_t count) {
  HIP_TRY(hipMalloc((void**)out, sizeof(Fr29Slot) * (size_t)count));
  hipLaunchKernelGGL(pow_table29_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, *out, base, first, count);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

// Coset tables for an arbitrary shift s (forward: s^i ; inverse: s^-i), i < 2^L, in both table forms.
// Used by the class decomposition of the quotient coset (prover.hip, multi-GPU): the quotient coset
// {g w_N^i} is the union of the size-n cosets (g w_N^j) H_n, so a class needs shift g w_N^j.
int ntt_coset_tables(Ctx* c, uint32_t L, const Fr& shift, bool inverse, NttCoset* out) {
  const Fr g = inverse ? shift.inv() : shift;
  const uint32_t ghi_n = L > (uint32_t)GLO_BITS ? (1u << (L - GLO_BITS)) : 1u;
  int rc;
  if ((rc = build_pow_table(c, &out->g_lo, g, Fr::one(), 1u << GLO_BITS))) return rc;
  if ((rc = build_pow_table(c, &out->g_hi, g.pow_u64(1ull << GLO_BITS), Fr::one(), ghi_n))) return rc;
  if ((rc = build_pow_table29(c, (Fr29Slot**)&out->g_lo29, g, Fr::one(), 1u << GLO_BITS))) return rc;
  if ((rc = build_pow_table29(c, (Fr29Slot**)&out->g_hi29, g.pow_u64(1ull << GLO_BITS), Fr::one(), ghi_n))) return rc;
  return PLONK_OK;
}
void ntt_coset_free(NttCoset* t) {
  (void)hipFree(t->g_lo); (void)hipFree(t->g_hi); (void)hipFree(t->g_lo29); (void)hipFree(t->g_hi29);
  *t = NttCoset{};
}

int ntt_tables(Ctx* c, uint32_t L, bool inverse, NttTables** out) {
  std::lock_guard<std::mutex> lk(c->table_mu);
  const uint32_t key = (L << 1) | (inverse ? 1u : 0u);
  auto it = c->ntt_tables.find(key);
  if (it != c->ntt_tables.end()) { *out = it->second; return PLONK_OK; }
  auto* tb = new NttTables();
  const Fr w = host_omega(L, inverse);
  const uint32_t lo_n = 1u << (L < (uint32_t)TWLO_BITS ? L : (uint32_t)TWLO_BITS);
  const uint32_t hi_n = L > (uint32_t)TWLO_BITS ? (1u << (L - TWLO_BITS)) : 1u;
  const Fr n_inv = Fr::from_u64(1ull << L).inv();
  int rc;
  if ((rc = build_pow_table(c, &tb->tw_lo, w, Fr::one(), lo_n))) return rc;
  if ((rc = build_pow_table(c, &tb->tw_hi, w.pow_u64(1ull << TWLO_BITS), Fr::one(), hi_n))) return rc;
  if (inverse) {
    if ((rc = build_pow_table(c, &tb->tw_lo_scaled, w, n_inv, lo_n))) return rc;
  }
  if ((rc = build_pow_table(c, &tb->w512, host_omega(9, inverse), Fr::one(), 256))) return rc;
  // coset tables: forward g^i ; inverse g^-i
  const Fr g = inverse ? fr_generator().inv() : fr_generator();
  const uint32_t ghi_n = L > (uint32_t)GLO_BITS ? (1u << (L - GLO_BITS)) : 1u;
  if ((rc = build_pow_table(c, &tb->g_lo, g, Fr::one(), 1u << GLO_BITS))) return rc;
  if ((rc = build_pow_table(c, &tb->g_hi, g.pow_u64(1ull << GLO_BITS), Fr::one(), ghi_n))) return rc;
  // the same tables as w * 2^261 / 29-bit limbs for the pass kernels
  if ((rc = build_pow_table29(c, (Fr29Slot**)&tb->tw_lo29, w, Fr::one(), lo_n))) return rc;
  if ((rc = build_pow_table29(c, (Fr29Slot**)&tb->tw_hi29, w.pow_u64(1ull << TWLO_BITS), Fr::one(), hi_n))) return rc;
  if (inverse) {
    if ((rc = build_pow_table29(c, (Fr29Slot**)&tb->tw_lo_scaled29, w, n_inv, lo_n))) return rc;
  }
  if ((rc = build_pow_table29(c, (Fr29Slot**)&tb->w512_29, host_omega(9, inverse), Fr::one(), 256))) return rc;
  if ((rc = build_pow_table29(c, (Fr29Slot**)&tb->g_lo29, g, Fr::one(), 1u << GLO_BITS))) return rc;
  if ((rc = build_pow_table29(c, (Fr29Slot**)&tb->g_hi29, g.pow_u64(1ull << GLO_BITS), Fr::one(), ghi_n))) return rc;
  // Direct inter-pass twiddles (one Fr product less per element in passes A and B, ~15 % of a transform's multiplications):
  // N + N / R1 slots of 48 B — 0.2 GB at 2^22, 0.8 GB at 2^24 per direction.  PLONK_NTT_DIRECT=0 keeps the two-level tables
  // (A/B runs); above 2^NTT_DIRECT_MAX_LOG, or when the tables would take more than 1/8 of the free memory, likewise.
  const bool direct_on = c->cfg.ntt_direct;
  if (direct_on && L > 10 && L <= (uint32_t)NTT_DIRECT_MAX_LOG) {
    int r[3], np;
    ntt_plan(L, r, &np);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t N = 1ull << L;
    if ((N + (N >> r[0])) * sizeof(Fr29Slot) <= free_b / 8) {
      // optional: on any failure the passes keep the two-level product
      if (build_pow_table29(c, (Fr29Slot**)&tb->tw_a29, w, inverse ? n_inv : Fr::one(), (uint32_t)N) != PLONK_OK) {
        (void)hipGetLastError();
        tb->tw_a29 = nullptr;
      }
      if (np == 3 && build_pow_table29(c, (Fr29Slot**)&tb->tw_b29, w.pow_u64(1ull << r[0]), Fr::one(), (uint32_t)(N >> r[0])) != PLONK_OK) {
        (void)hipGetLastError();
        tb->tw_b29 = nullptr;
      }
    }
  }
  tb->n_inv = n_inv;
  // The tables were filled by kernels queued on the CURRENT stream — which is the low-priority side stream when a size is first
  // needed inside a SideScope (the wire inverse transforms of a first proof at <= 2^18 gates) — and every later user finds them in
  // the map, on whatever stream it runs: without a wait here the main stream's next transform of that size races the fill.
  // Found in round 6 by running four provers on one GPU at once (tests/test_gpu_msm_variants.py children in parallel): with the
  // side stream starved, the first proof of a fresh context read an unwritten twiddle table about once in eight runs and
  // returned CircuitUnsatisfied.  One wait per (size, direction) per context, at creation.
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->ntt_tables[key] = tb;
  *out = tb;
  return PLONK_OK;
}

template <int RLOG, int ELOG>
static void launch_pass(Ctx* c, const NttPass& p, bool transpose, uint64_t cols) {
  constexpr int TILE_LOG = NTT_THREAD_BITS + ELOG;
  constexpr size_t smem = (size_t)Fr29::N * (1 << TILE_LOG) * sizeof(uint32_t);
  const uint32_t nblocks = (uint32_t)(cols >> (TILE_LOG - RLOG));   // a workgroup owns 2^(TILE_LOG - RLOG) columns of R rows
  if (transpose) {
    smem_opt_in(c, (const void*)ntt_pass_kernel<RLOG, true, ELOG>, smem);
    hipLaunchKernelGGL((ntt_pass_kernel<RLOG, true, ELOG>), dim3(nblocks), dim3(NTT_THREADS), smem, c->stream, p);
  } else {
    smem_opt_in(c, (const void*)ntt_pass_kernel<RLOG, false, ELOG>, smem);
    hipLaunchKernelGGL((ntt_pass_kernel<RLOG, false, ELOG>), dim3(nblocks), dim3(NTT_THREADS), smem, c->stream, p);
  }
}

// Elements per lane of the pass kernels (log2): PLONK_NTT_ELOG=2|3 forces either, else the caller's hint (prover.hip), else
// the default; a radix of 2^9 always runs with 8 elements (a 1024-element tile would be two columns wide: 64-byte runs).
static int ntt_elog(const Ctx* c, int rlog) {
  const int forced = c->cfg.ntt_elog;   // plonk_gpu_config.ntt_elements_log2 / PLONK_NTT_ELOG
  if (rlog >= 9) return 3;
  if (forced) return forced;
  return c->ntt_elog_hint == 2 || c->ntt_elog_hint == 3 ? c->ntt_elog_hint : NTT_ELOG_DEFAULT;
}

// cols: columns of the [R = 2^rlog rows][cols] view the pass transforms (N / R)
static int launch_pass_rt(Ctx* c, int rlog, const NttPass& p, bool transpose, uint64_t cols) {
  const int elog = ntt_elog(c, rlog);
  switch (rlog * 4 + elog) {
    // ntt_plan() only produces radices 5..9 for N >= 2^11
    case 5 * 4 + 3: launch_pass<5, 3>(c, p, transpose, cols); break;
    case 6 * 4 + 3: launch_pass<6, 3>(c, p, transpose, cols); break;
    case 7 * 4 + 3: launch_pass<7, 3>(c, p, transpose, cols); break;
    case 8 * 4 + 3: launch_pass<8, 3>(c, p, transpose, cols); break;
    case 9 * 4 + 3: launch_pass<9, 3>(c, p, transpose, cols); break;
    case 5 * 4 + 2: launch_pass<5, 2>(c, p, transpose, cols); break;
    case 6 * 4 + 2: launch_pass<6, 2>(c, p, transpose, cols); break;
    case 7 * 4 + 2: launch_pass<7, 2>(c, p, transpose, cols); break;
    case 8 * 4 + 2: launch_pass<8, 2>(c, p, transpose, cols); break;
    default: return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  }
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

// Device-resident transform: src (in_len valid elements) -> dst (N elements).
// src == dst is allowed.  `tmp` must hold N elements (N > 1024 only).
int ntt_device(Ctx* c, const Fr* src, Fr* dst, Fr* tmp, uint32_t L, bool inverse, bool coset, uint64_t in_len,
               const NttCoset* shift) {
  if (L >= 28) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);   // 3 passes of <= 2^9; reference limit is 2^32 (domain.rs:132)
  const uint64_t N = 1ull << L;
  if (in_len > N) in_len = N;          // Vec::resize truncation, domain.rs:174
  if (in_len == 0) {                   // transform of the zero vector (every variant is linear); the pass kernels read src[0]
    HIP_TRY(hipMemsetAsync(dst, 0, sizeof(Fr) * N, c->stream));
    return PLONK_OK;
  }
  NttTables* tb;
  int rc = ntt_tables(c, L, inverse, &tb);
  if (rc) return rc;
  int r[3], np;
  ntt_plan(L, r, &np);
  // coset tables: the generator's (reference domain.rs:198-232), or the caller's for another shift
  const Fr* g_lo = shift ? shift->g_lo : tb->g_lo;
  const Fr* g_hi = shift ? shift->g_hi : tb->g_hi;
  const Fr29Slot* g_lo29 = (const Fr29Slot*)(shift ? shift->g_lo29 : tb->g_lo29);
  const Fr29Slot* g_hi29 = (const Fr29Slot*)(shift ? shift->g_hi29 : tb->g_hi29);
  const int post = inverse ? (coset ? 2 : 1) : 0;
  if (np == 1) {
    NttSmall s{};
    s.src = src; s.dst = dst; s.logN = L; s.in_len = in_len;
    s.pre_coset = (coset && !inverse) ? 1 : 0;
    s.post_mode = post; s.scale = tb->n_inv;
    s.tw_lo = tb->tw_lo; s.g_lo = g_lo; s.g_hi = g_hi;
    hipLaunchKernelGGL(ntt_small_kernel, dim3(1), dim3(NTT_THREADS), 0, c->stream, s);
    HIP_TRY(hipGetLastError());
    return PLONK_OK;
  }
  const uint64_t R1 = 1ull << r[0];
  const int rl = r[np - 1];
  // ---- pass A : src -> tmp (transposed)
  {
    NttPass p{};
    p.src = src; p.dst = tmp; p.logN = L;
    p.in_rs = N >> r[0]; p.in_hshift = 63; p.in_hs = 0;
    p.out_rs = 1; p.out_hshift = (uint32_t)rl; p.out_hs = R1; p.out_ls = N >> rl;
    p.tw_mode = 1; p.tw_shr = 0;
    p.tw_lo = (const Fr29Slot*)(inverse ? tb->tw_lo_scaled29 : tb->tw_lo29);   // n^-1 folded into pass A's twiddles
    p.tw_hi = (const Fr29Slot*)tb->tw_hi29;
    p.tw_direct = (const Fr29Slot*)tb->tw_a29;
    p.in_len = in_len; p.pre_coset = (coset && !inverse) ? 1 : 0;
    p.post_mode = 0; p.g_lo = g_lo29; p.g_hi = g_hi29;
    p.w512 = (const Fr29Slot*)tb->w512_29;
    if ((rc = launch_pass_rt(c, r[0], p, true, N >> r[0]))) return rc;
  }
  // ---- pass B : tmp in place
  if (np == 3) {
    NttPass p{};
    p.src = tmp; p.dst = tmp; p.logN = L;
    p.in_rs = R1; p.in_hshift = (uint32_t)r[0]; p.in_hs = R1 << r[1];
    p.out_rs = R1; p.out_hshift = (uint32_t)r[0]; p.out_hs = R1 << r[1]; p.out_ls = 1;
    p.tw_mode = 1; p.tw_shr = (uint32_t)r[0];
    p.tw_lo = (const Fr29Slot*)tb->tw_lo29; p.tw_hi = (const Fr29Slot*)tb->tw_hi29;
    p.tw_direct = (const Fr29Slot*)tb->tw_b29;
    p.in_len = N; p.pre_coset = 0; p.post_mode = 0; p.w512 = (const Fr29Slot*)tb->w512_29;
    p.g_lo = g_lo29; p.g_hi = g_hi29;
    if ((rc = launch_pass_rt(c, r[1], p, false, N >> r[1]))) return rc;
  }
  // ---- pass C : tmp -> dst
  {
    NttPass p{};
    p.src = tmp; p.dst = dst; p.logN = L;
    p.in_rs = N >> rl; p.in_hshift = 63; p.in_hs = 0;
    p.out_rs = N >> rl; p.out_hshift = 63; p.out_hs = 0; p.out_ls = 1;
    p.tw_mode = 0; p.tw_shr = 0; p.tw_lo = (const Fr29Slot*)tb->tw_lo29; p.tw_hi = (const Fr29Slot*)tb->tw_hi29;
    p.in_len = N; p.pre_coset = 0;
    p.post_mode = (inverse && coset) ? 2 : 0;   // n^-1 already folded in pass A
    p.scale = Fr29::zero(); p.g_lo = g_lo29; p.g_hi = g_hi29;
    p.w512 = (const Fr29Slot*)tb->w512_29;
    if ((rc = launch_pass_rt(c, rl, p, false, N >> rl))) return rc;
  }
  return PLONK_OK;
}

}  // namespace plonk
