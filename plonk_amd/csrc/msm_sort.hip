// Bucket grouping for the Pippenger MSM (msm.hip): from the scalars of a commitment group straight to
// table entries grouped by bucket — a hand-written two-level counting sort for gfx950, no library
// primitive.  Order inside a bucket is irrelevant to a sum, which is what makes this cheaper than a
// general (stable) radix sort:
//
//   msm_hist_kernel       scalars -> digits (msm_recode.cuh: signed 16-bit windows for window tables, width-17 NAF
//                         for bit-position tables) -> histogram of the COARSE bin (top 11 of the 15 bucket bits) in
//                         LDS, flushed with one global atomic per (workgroup, bin).  Zero digits are dropped here and
//                         never touched again.
//   msm_coarse_scan       2048 counts -> coarse offsets (one workgroup per commitment).
//   msm_partition_kernel  scalars again (32 B per term instead of a 6-byte key/value pair per window):
//                         a workgroup owns a tile of 2048 scalars = 32768 entries, ranks them per coarse
//                         bin with LDS atomics, reserves its run in every bin with one global atomic,
//                         regroups the tile in LDS (128 KiB of the CU's 160 KiB) and writes runs of
//                         ~16 consecutive words.  The word carries what the second level needs: fine bucket (4 bits) |
//                         sign | table index — 32 bits while the tables have at most 2^27 entries, 64 bits beyond
//                         (bit-position tables of more than 2^19 points; the LDS staging keeps a 24-bit tile-local form
//                         either way: fine | sign | row | index inside the tile).
//   msm_fine_kernel       one workgroup per coarse bin (~8192 entries, held in registers between the two
//                         steps): histogram of the 16 fine buckets, then scatter into the final entry array;
//                         writes the bucket offsets on the way, so no pass over 32768 counters is needed.
//   msm_slices_kernel     bucket offsets -> slice offsets (prefix sum of ceil(count / KSL)).
//
// HBM traffic per commitment of m terms: 2 x 32m (scalars, read twice) + 64m written and read back
// (coarse-partitioned words) + 64m written (entries) = 256m bytes, against 560m for digits + key/value
// pairs + a two-pass onesweep.  A bucket that attracts a large share of the digits (equal
// coefficients) only makes one workgroup loop longer; nothing overflows.
#include "plonk_internal.hpp"
#include "fr29.cuh"

namespace plonk {
namespace PLONK_MSM_NS {   // compiled once per bucket count (plonk_internal.hpp)

static constexpr uint32_t FINE_BITS = MSM_NB_BITS - 11;          // 4 fine bits for 2^15 buckets, 8 for 2^19
static constexpr uint32_t COARSE = MSM_NB >> FINE_BITS;          // 2048 coarse bins
#ifndef PLONK_SORT_TILE
#define PLONK_SORT_TILE 2048
#endif
static constexpr uint32_t TILE = PLONK_SORT_TILE;                // scalars per workgroup of the partition pass
static constexpr uint32_t SORT_T = 1024;                         // threads per workgroup
static constexpr uint32_t PER_T = TILE / SORT_T;                 // scalars per thread (partition)
#ifndef PLONK_HIST_PER
#define PLONK_HIST_PER 2
#endif
static constexpr uint32_t HIST_PER = PLONK_HIST_PER;             // scalars per thread of the histogram pass
static constexpr uint32_t HIST_TILE = SORT_T * HIST_PER;
static_assert(COARSE == 2 * SORT_T, "scan / reservation loops assume two coarse bins per thread");
// Intermediate (coarse-partitioned) word.  Narrow: fine (4) | sign (1) | table index (27).  Wide: fine in the upper
// half, the lower half IS the final entry (sign << 31 | 31-bit table index).
static constexpr uint32_t IDX_BITS = 31 - FINE_BITS;             // 27 (2^15 buckets) / 23 (2^19)
static constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1;
static_assert(IDX_BITS + 1 + FINE_BITS == 32, "narrow intermediate word layout");
template <class WordT> struct SortWord;
template <> struct SortWord<uint32_t> {
  __device__ static __forceinline__ uint32_t make(uint32_t fine, uint32_t sign, uint64_t idx) { return (fine << (IDX_BITS + 1)) | (sign << IDX_BITS) | (uint32_t)idx; }
  __device__ static __forceinline__ uint32_t fine(uint32_t w) { return w >> (IDX_BITS + 1); }
  __device__ static __forceinline__ uint32_t entry(uint32_t w) { return (w & IDX_MASK) | (((w >> IDX_BITS) & 1u) << 31); }
};
template <> struct SortWord<uint64_t> {
  __device__ static __forceinline__ uint64_t make(uint32_t fine, uint32_t sign, uint64_t idx) { return ((uint64_t)fine << 32) | ((uint64_t)sign << 31) | idx; }
  __device__ static __forceinline__ uint32_t fine(uint64_t w) { return (uint32_t)(w >> 32); }
  __device__ static __forceinline__ uint32_t entry(uint64_t w) { return (uint32_t)w; }
};
// The coarse-partitioned words in HBM.  Narrow: one u32 plane.  Wide (register form fine << 32 | entry): a u32 plane with
// the entry and a BYTE plane with the fine bucket — 5 bytes per word out and back instead of 8, and the histogram passes
// read only the byte plane.  Both planes live in MsmWork::tmp_words: [KB][W * cap] u32, then [KB][W * cap] u8.
template <class WordT> struct TmpPlanes;
template <> struct TmpPlanes<uint32_t> {
  uint32_t* lo;
  __device__ __forceinline__ TmpPlanes(void* all, int kb, uint64_t words) : lo((uint32_t*)all + (uint64_t)kb * words) {}
  __device__ __forceinline__ uint32_t ld(uint32_t j) const { return lo[j]; }
  __device__ __forceinline__ uint32_t ld_fine(uint32_t j) const { return SortWord<uint32_t>::fine(lo[j]); }
  __device__ __forceinline__ void st(uint32_t j, uint32_t w) const { lo[j] = w; }
};
template <> struct TmpPlanes<uint64_t> {
  uint32_t* lo;
  uint8_t* hi;
  __device__ __forceinline__ TmpPlanes(void* all, int kb, uint64_t words)
      : lo((uint32_t*)all + (uint64_t)kb * words), hi((uint8_t*)((uint32_t*)all + (uint64_t)MSM_MAX_BATCH * words) + (uint64_t)kb * words) {}
  __device__ __forceinline__ uint64_t ld(uint32_t j) const { return ((uint64_t)hi[j] << 32) | lo[j]; }
  __device__ __forceinline__ uint32_t ld_fine(uint32_t j) const { return hi[j]; }
  __device__ __forceinline__ void st(uint32_t j, uint64_t w) const { lo[j] = (uint32_t)w; hi[j] = (uint8_t)(w >> 32); }
};
// tile-local form staged in LDS by msm_partition: fine (4) | sign (1) | table row (8) | scalar inside the tile (11)
static_assert(TILE <= 2048, "tile-local word: 11 bits of scalar index");

__device__ __forceinline__ Fr ld_scalar(const Fr* p);
// scalar i of commitment kb: the main array below the split, the tail array above
__device__ __forceinline__ Fr ld_scalar_at(const MsmBatch& bt, int kb, uint64_t i) {
  return i < bt.split[kb] ? ld_scalar(bt.scalars[kb] + i) : ld_scalar(bt.tail[kb] + (i - bt.split[kb]));
}
__device__ __forceinline__ Fr ld_scalar(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}

// Montgomery scalar -> canonical integer: (x * 2^256) * 32 / 2^261 = x in reduced radix, exact canonicalisation
__device__ __forceinline__ Fr scalar_canonical(const Fr& mont) {
  Fr29 c32 = Fr29::zero();
  c32.l[0] = 32;
  return Fr29::mul(Fr29::from_fr(mont), c32).to_fr();
}

// ---- level 1a: coarse histogram -------------------------------------------------------------------
// MODE: how the scalars are recoded (msm_recode.cuh) — 0 signed 16-bit windows (window tables, 2^15 buckets only), 1 width-w NAF
// (a table row per bit position), 2 even-position digits (a row for every second bit position, round 4)
static constexpr uint32_t MSM_EVEN_WIDTH = (MSM_NB_BITS + 1) & ~1u;   // even-position digits: |d| <= 2^(width - 1) <= 2^NB_BITS, bucket = |d| - 1
template <int MODE>
__global__ void __launch_bounds__(SORT_T) msm_hist_kernel(MsmBatch bt, uint32_t* __restrict__ coarse_cnt_all) {
  constexpr bool BITPOS = MODE != 0;                   // the scalar is parked in LDS for the run-time bit positions of modes 1 and 2
  __shared__ uint32_t hist[COARSE];
  __shared__ uint32_t park[BITPOS ? 9 * SORT_T : 1];   // bit-position recoding: the canonical scalar, limb-major, + a zero limb (StridedLimbs)
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint64_t m = bt.m[kb];
  const uint64_t base = (uint64_t)blockIdx.x * HIST_TILE;
  if (base >= m) return;
  const uint32_t t = threadIdx.x;
  hist[t] = 0;
  hist[t + SORT_T] = 0;
  __syncthreads();
  Fr raw[HIST_PER];   // all loads of the thread in flight before the first conversion
#pragma unroll
  for (uint32_t k = 0; k < HIST_PER; ++k) {
    const uint64_t i = base + t + (uint64_t)k * SORT_T;
    if (i < m) raw[k] = ld_scalar_at(bt, kb, i);
  }
#pragma unroll
  for (uint32_t k = 0; k < HIST_PER; ++k) {
    const uint64_t i = base + t + (uint64_t)k * SORT_T;
    if (i < m) {
      const Fr s = scalar_canonical(raw[k]);
      auto count = [&](int, uint32_t, uint32_t bucket, uint32_t) { atomicAdd(&hist[bucket >> FINE_BITS], 1u); };
      if (BITPOS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) park[j * SORT_T + t] = s.l[j];   // read back by this lane only: no barrier
        park[8 * SORT_T + t] = 0;
        if (MODE == 2) for_each_digit_even<MSM_EVEN_WIDTH>(StridedLimbs{park + t, SORT_T}, count);
        else for_each_digit_naf<MSM_NAF_WIDTH>(StridedLimbs{park + t, SORT_T}, count);
      } else if constexpr (MSM_NB_BITS == 15) {
        for_each_digit_window(s, count);
      }
    }
  }
  __syncthreads();
  uint32_t* __restrict__ cnt = coarse_cnt_all + (uint64_t)kb * COARSE;
  const uint32_t a = hist[t], b = hist[t + SORT_T];
  if (a) atomicAdd(&cnt[t], a);
  if (b) atomicAdd(&cnt[t + SORT_T], b);
}

// exclusive scan of `v` over the SORT_T threads of a workgroup (Hillis-Steele in LDS); returns the
// exclusive prefix of this thread's value, *total = sum of all
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* sh /* SORT_T */, uint32_t* total) {
  const uint32_t t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (uint32_t d = 1; d < SORT_T; d <<= 1) {
    const uint32_t x = t >= d ? sh[t - d] : 0;
    __syncthreads();
    sh[t] += x;
    __syncthreads();
  }
  const uint32_t incl = sh[t];
  *total = sh[SORT_T - 1];
  __syncthreads();
  return incl - v;
}

// exclusive prefix of the 2^FINE_BITS counters cnt[] into ex[] (sc: scratch of the same size; *total = their sum); every
// thread of the workgroup calls it.  The serial loop of one thread it replaces was free for 16 fine buckets and 5-10 us —
// with a global atomic per bucket in msm_big_scatter, 160 us — for the 256 of the 2^19-bucket variant.
__device__ __forceinline__ void fine_exclusive_scan(const uint32_t* cnt, uint32_t* ex, uint32_t* sc, uint32_t* total) {
  constexpr uint32_t NF = 1u << FINE_BITS;
  const uint32_t t = threadIdx.x;
  const uint32_t v = t < NF ? cnt[t] : 0u;
  if (t < NF) sc[t] = v;
  __syncthreads();
  for (uint32_t d = 1; d < NF; d <<= 1) {
    const uint32_t x = (t < NF && t >= d) ? sc[t - d] : 0u;
    __syncthreads();
    if (t < NF) sc[t] += x;
    __syncthreads();
  }
  if (t < NF) ex[t] = sc[t] - v;
  *total = sc[NF - 1];
  __syncthreads();
}

// ---- level 1b: coarse offsets; also clears the run cursors of the partition pass ------------------
// Skewed digits (many equal or small scalars: bits, quads, range accumulators of a real witness) put a large share
// of the entries into a few coarse bins.  A bin above BIG_LIMIT words is not left to one workgroup: it is cut into
// chunks of BIG_CHUNK words that msm_big_hist / msm_big_scatter process in parallel (chunk prefix in big_off).
static constexpr uint32_t BIG_LIMIT = 1u << 16;
static constexpr uint32_t BIG_CHUNK = 1u << 14;
__device__ __forceinline__ uint32_t big_chunks(uint32_t cnt) { return cnt > BIG_LIMIT ? (cnt + BIG_CHUNK - 1) / BIG_CHUNK : 0u; }

__global__ void __launch_bounds__(SORT_T) msm_coarse_scan_kernel(const uint32_t* __restrict__ coarse_cnt_all,
                                                                 uint32_t* __restrict__ coarse_off_all,
                                                                 uint32_t* __restrict__ coarse_cur_all,
                                                                 uint32_t* __restrict__ big_off_all, int kb0) {
  __shared__ uint32_t sh[SORT_T];
  const uint32_t* __restrict__ cnt = coarse_cnt_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * COARSE;
  uint32_t* __restrict__ off = coarse_off_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * (COARSE + 1);
  uint32_t* __restrict__ cur = coarse_cur_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * COARSE;
  uint32_t* __restrict__ big = big_off_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * (COARSE + 1);
  const uint32_t t = threadIdx.x;
  const uint32_t c0 = cnt[2 * t], c1 = cnt[2 * t + 1];
  uint32_t total;
  const uint32_t ex = block_exclusive_scan(c0 + c1, sh, &total);
  off[2 * t] = ex;
  off[2 * t + 1] = ex + c0;
  cur[2 * t] = 0;
  cur[2 * t + 1] = 0;
  if (t == SORT_T - 1) off[COARSE] = total;
  const uint32_t b0 = big_chunks(c0), b1 = big_chunks(c1);
  uint32_t btotal;
  const uint32_t bex = block_exclusive_scan(b0 + b1, sh, &btotal);
  big[2 * t] = bex;
  big[2 * t + 1] = bex + b0;
  if (t == SORT_T - 1) big[COARSE] = btotal;
}

// ---- level 1c: partition into the coarse bins -----------------------------------------------------
// dynamic LDS: stage[TILE * MSM_W] words, then hist / loff / gbase [COARSE] each
static constexpr size_t PARTITION_LDS = ((size_t)TILE * MSM_W + 3 * COARSE) * sizeof(uint32_t);

template <int MODE, class WordT>
__global__ void __launch_bounds__(SORT_T) msm_partition_kernel(MsmBatch bt, uint64_t srs_n,
                                                               const uint32_t* __restrict__ coarse_off_all,
                                                               uint32_t* __restrict__ coarse_cur_all,
                                                               void* __restrict__ tmp_all) {
  constexpr bool BITPOS = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* stage = lds;                          // TILE * MSM_W
  uint32_t* hist = lds + TILE * MSM_W;            // COARSE: entries of this tile per bin
  uint32_t* loff = hist + COARSE;                 // COARSE: start of the bin's run inside `stage`
  uint32_t* gbase = loff + COARSE;                // COARSE: start of the run in the global array
  __shared__ uint32_t sh[SORT_T];
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint64_t m = bt.m[kb];
  const uint64_t base = (uint64_t)blockIdx.x * TILE;
  if (base >= m) return;
  const uint32_t t = threadIdx.x;
  hist[t] = 0;
  hist[t + SORT_T] = 0;
  __syncthreads();
  // words and (bin << 16 | rank) of this thread's entries stay in registers until the runs are known
  uint32_t word[PER_T][MSM_W], where[PER_T][MSM_W];
#pragma unroll
  for (uint32_t k = 0; k < PER_T; ++k) {
#pragma unroll
    for (int w = 0; w < MSM_W; ++w) where[k][w] = 0xffffffffu;
    const uint64_t i = base + t + (uint64_t)k * SORT_T;
    if (i < m) {
      const Fr s = scalar_canonical(ld_scalar_at(bt, kb, i));
      auto put = [&](int w, uint32_t row, uint32_t bucket, uint32_t sign) {   // w = digit slot (static after unrolling)
        const uint32_t bin = bucket >> FINE_BITS;
        const uint32_t rank = atomicAdd(&hist[bin], 1u);          // < TILE * MSM_W = 2^15
        where[k][w] = (bin << 16) | rank;
        word[k][w] = ((bucket & ((1u << FINE_BITS) - 1)) << 20) | (sign << 19) | (row << 11) | (t + k * SORT_T);
      };
      if (BITPOS) {   // the canonical scalar parked in the (still unused) staging area, limb-major; only this lane reads it back
        uint32_t* park = stage + k * (9 * SORT_T);
#pragma unroll
        for (int j = 0; j < 8; ++j) park[j * SORT_T + t] = s.l[j];
        park[8 * SORT_T + t] = 0;
        if (MODE == 2) for_each_digit_even<MSM_EVEN_WIDTH>(StridedLimbs{park + t, SORT_T}, put);
        else for_each_digit_naf<MSM_NAF_WIDTH>(StridedLimbs{park + t, SORT_T}, put);
      } else if constexpr (MSM_NB_BITS == 15) {
        for_each_digit_window(s, put);
      }
    }
  }
  __syncthreads();   // (also: every parked scalar has been read before the staging area is written below)
  {
    const uint32_t c0 = hist[2 * t], c1 = hist[2 * t + 1];
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(c0 + c1, sh, &total);
    loff[2 * t] = ex;
    loff[2 * t + 1] = ex + c0;
    const uint32_t* __restrict__ coff = coarse_off_all + (uint64_t)kb * (COARSE + 1);
    uint32_t* __restrict__ cur = coarse_cur_all + (uint64_t)kb * COARSE;
    gbase[2 * t] = c0 ? coff[2 * t] + atomicAdd(&cur[2 * t], c0) : 0;
    gbase[2 * t + 1] = c1 ? coff[2 * t + 1] + atomicAdd(&cur[2 * t + 1], c1) : 0;
  }
  __syncthreads();
  const TmpPlanes<WordT> tmp(tmp_all, kb, (uint64_t)MSM_W * bt.cap_m);
#pragma unroll
  for (uint32_t k = 0; k < PER_T; ++k)
#pragma unroll
    for (int w = 0; w < MSM_W; ++w)
      if (where[k][w] != 0xffffffffu) stage[loff[where[k][w] >> 16] + (where[k][w] & 0xffffu)] = word[k][w];
  __syncthreads();
  // write the runs: 16 lanes per bin, 64 bins per sweep of the workgroup
  const uint32_t sub = t & 15;
  for (uint32_t bin = t >> 4; bin < COARSE; bin += SORT_T / 16) {
    const uint32_t cnt = hist[bin], lo = loff[bin], gb = gbase[bin];
    for (uint32_t j = sub; j < cnt; j += 16) {
      const uint32_t lw = stage[lo + j];          // tile-local word -> global word: table index = row * points + scalar
      tmp.st(gb + j, SortWord<WordT>::make(lw >> 20, (lw >> 19) & 1u, (uint64_t)((lw >> 11) & 0xffu) * srs_n + base + (lw & 0x7ffu)));
    }
  }
}

#if PLONK_MSM_NB_BITS >= 19   // (13 digit slots need digits at least 20 bits apart: the 2^19-bucket build only)
// ---- level 1c, round 5 variant: two partition workgroups per CU -----------------------------------------------------
// The partition pass above stages a tile of 2048 scalars x 16 digit slots: 128 KiB of the CU's 160 KiB of LDS, i.e. ONE
// workgroup per CU — while it scans its histogram (20 barriers) or drains its runs to HBM nothing else on that CU issues.
// Over 2^19 buckets a scalar has at most 13 digits (width 21: ceil(256 / 21); half-density width 20: 13), so a tile of 1024
// scalars needs 1024 x 13 words = 52 KiB + the three 8 KiB bin arrays = 76 KiB: TWO workgroups of 512 threads per CU (four
// waves per SIMD at <= 128 VGPRs), one in its memory phase while the other computes.  The price: a tile contributes ~6
// words to a coarse bin instead of ~12, so the runs written to HBM are half as long.  Selected by PLONK_MSM_SORT13=1 (A/B;
// the measurement decides the default: profiles/r05/SUMMARY.md section 2).  Same words, same order-insensitive result.
static constexpr uint32_t P2_TILE = 1024, P2_T = 512, P2_W = 13, P2_PER = P2_TILE / P2_T, P2_BPT = COARSE / P2_T;
static constexpr size_t PARTITION2_LDS = ((size_t)P2_TILE * P2_W + 3 * COARSE) * sizeof(uint32_t);
static_assert(P2_PER * 9 * P2_T <= P2_TILE * P2_W, "the parked scalars fit the staging area");
// consecutive digits of either recoding are at least their width apart, so a 256-bit scalar has at most ceil(256 / width)
// of them: the 13 slots can never overflow (the `w < P2_W` test in `put` only keeps the unrolled indices static)
static_assert((MSM_NAF_WIDTH) * P2_W >= 256 && (MSM_EVEN_WIDTH) * P2_W >= 256, "13 digit slots cover a scalar");

__device__ __forceinline__ uint32_t block_exclusive_scan_p2(uint32_t v, uint32_t* sh /* P2_T */, uint32_t* total) {
  const uint32_t t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (uint32_t d = 1; d < P2_T; d <<= 1) {
    const uint32_t x = t >= d ? sh[t - d] : 0;
    __syncthreads();
    sh[t] += x;
    __syncthreads();
  }
  const uint32_t incl = sh[t];
  *total = sh[P2_T - 1];
  __syncthreads();
  return incl - v;
}

template <int MODE, class WordT>
__global__ void __launch_bounds__(P2_T, 4) msm_partition2_kernel(MsmBatch bt, uint64_t srs_n,
                                                                  const uint32_t* __restrict__ coarse_off_all,
                                                                  uint32_t* __restrict__ coarse_cur_all,
                                                                  void* __restrict__ tmp_all) {
  static_assert(MODE != 0, "bit-position / half-density recodings only");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* stage = lds;                          // P2_TILE * P2_W
  uint32_t* hist = lds + P2_TILE * P2_W;          // COARSE
  uint32_t* loff = hist + COARSE;
  uint32_t* gbase = loff + COARSE;
  __shared__ uint32_t sh[P2_T];
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint64_t m = bt.m[kb];
  const uint64_t base = (uint64_t)blockIdx.x * P2_TILE;
  if (base >= m) return;
  const uint32_t t = threadIdx.x;
#pragma unroll
  for (uint32_t j = 0; j < P2_BPT; ++j) hist[t + j * P2_T] = 0;
  __syncthreads();
  uint32_t word[P2_PER][P2_W], where[P2_PER][P2_W];
#pragma unroll
  for (uint32_t k = 0; k < P2_PER; ++k) {
#pragma unroll
    for (uint32_t w = 0; w < P2_W; ++w) where[k][w] = 0xffffffffu;
    const uint64_t i = base + t + (uint64_t)k * P2_T;
    if (i < m) {
      const Fr s = scalar_canonical(ld_scalar_at(bt, kb, i));
      auto put = [&](int w, uint32_t row, uint32_t bucket, uint32_t sign) {   // w = digit slot (static after unrolling), < 13 by the digit widths
        if (w < (int)P2_W) {
          const uint32_t bin = bucket >> FINE_BITS;
          const uint32_t rank = atomicAdd(&hist[bin], 1u);          // < P2_TILE * P2_W < 2^14
          where[k][w] = (bin << 16) | rank;
          word[k][w] = ((bucket & ((1u << FINE_BITS) - 1)) << 20) | (sign << 19) | (row << 11) | (t + k * P2_T);
        }
      };
      uint32_t* park = stage + k * (9 * P2_T);   // the canonical scalar, limb-major; only this lane reads it back
#pragma unroll
      for (int j = 0; j < 8; ++j) park[j * P2_T + t] = s.l[j];
      park[8 * P2_T + t] = 0;
      if (MODE == 2) for_each_digit_even<MSM_EVEN_WIDTH>(StridedLimbs{park + t, P2_T}, put);
      else for_each_digit_naf<MSM_NAF_WIDTH>(StridedLimbs{park + t, P2_T}, put);
    }
  }
  __syncthreads();   // (also: every parked scalar has been read before the staging area is written below)
  {
    uint32_t cnt[P2_BPT], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < P2_BPT; ++j) { cnt[j] = hist[P2_BPT * t + j]; sum += cnt[j]; }
    uint32_t total;
    uint32_t run = block_exclusive_scan_p2(sum, sh, &total);
    const uint32_t* __restrict__ coff = coarse_off_all + (uint64_t)kb * (COARSE + 1);
    uint32_t* __restrict__ cur = coarse_cur_all + (uint64_t)kb * COARSE;
#pragma unroll
    for (uint32_t j = 0; j < P2_BPT; ++j) {
      const uint32_t bin = P2_BPT * t + j;
      loff[bin] = run;
      run += cnt[j];
      gbase[bin] = cnt[j] ? coff[bin] + atomicAdd(&cur[bin], cnt[j]) : 0;
    }
  }
  __syncthreads();
  const TmpPlanes<WordT> tmp(tmp_all, kb, (uint64_t)MSM_W * bt.cap_m);
#pragma unroll
  for (uint32_t k = 0; k < P2_PER; ++k)
#pragma unroll
    for (uint32_t w = 0; w < P2_W; ++w)
      if (where[k][w] != 0xffffffffu) stage[loff[where[k][w] >> 16] + (where[k][w] & 0xffffu)] = word[k][w];
  __syncthreads();
  // write the runs: 8 lanes per bin (a run is ~6 words), 64 bins per sweep of the workgroup
  const uint32_t sub = t & 7;
  for (uint32_t bin = t >> 3; bin < COARSE; bin += P2_T / 8) {
    const uint32_t cnt = hist[bin], lo = loff[bin], gb = gbase[bin];
    for (uint32_t j = sub; j < cnt; j += 8) {
      const uint32_t lw = stage[lo + j];
      tmp.st(gb + j, SortWord<WordT>::make(lw >> 20, (lw >> 19) & 1u, (uint64_t)((lw >> 11) & 0xffu) * srs_n + base + (lw & 0x7ffu)));
    }
  }
}
#endif

// ---- level 2: fine buckets inside a coarse bin ----------------------------------------------------
#ifndef PLONK_FINE_T
#define PLONK_FINE_T 512
#endif
#ifndef PLONK_FINE_CACHE
#define PLONK_FINE_CACHE 20
#endif
static constexpr uint32_t FINE_T = PLONK_FINE_T;
static constexpr uint32_t FINE_CACHE = PLONK_FINE_CACHE;   // words per thread kept in registers between the two passes: 512 x 20 covers a bin of
                                                           // 10240 words (mean 8192 at 2^20 terms), anything beyond is re-read; A/B r02: -0.25 ms per proof
template <class WordT>
__global__ void __launch_bounds__(FINE_T) msm_fine_kernel(MsmBatch bt, const uint32_t* __restrict__ coarse_off_all,
                                                          void* __restrict__ tmp_all,
                                                          uint32_t* __restrict__ entries_all,
                                                          uint32_t* __restrict__ offsets_all) {
  __shared__ uint32_t cnt[1u << FINE_BITS], start[1u << FINE_BITS], cur[1u << FINE_BITS], scan_tmp[1u << FINE_BITS];
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint32_t bin = blockIdx.x, t = threadIdx.x;
  const uint32_t* __restrict__ coff = coarse_off_all + (uint64_t)kb * (COARSE + 1);
  const TmpPlanes<WordT> tmp(tmp_all, kb, (uint64_t)MSM_W * bt.cap_m);
  uint32_t* __restrict__ entries = entries_all + (uint64_t)kb * MSM_W * bt.cap_m;
  uint32_t* __restrict__ offsets = offsets_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t beg = coff[bin], end = coff[bin + 1];
  if (end - beg > BIG_LIMIT) return;   // oversized bin: msm_big_hist / msm_big_scatter
  if (t < (1u << FINE_BITS)) { cnt[t] = 0; cur[t] = 0; }
  __syncthreads();
  WordT cache[FINE_CACHE > 0 ? FINE_CACHE : 1];
#pragma unroll
  for (uint32_t r = 0; r < FINE_CACHE; ++r) {
    const uint32_t j = beg + t + r * FINE_T;
    cache[r] = j < end ? tmp.ld(j) : (WordT)0;
  }
#pragma unroll
  for (uint32_t r = 0; r < FINE_CACHE; ++r)
    if (beg + t + r * FINE_T < end) atomicAdd(&cnt[SortWord<WordT>::fine(cache[r])], 1u);
  for (uint32_t j = beg + t + FINE_CACHE * FINE_T; j < end; j += FINE_T) atomicAdd(&cnt[tmp.ld_fine(j)], 1u);
  __syncthreads();
  {
    uint32_t total;
    fine_exclusive_scan(cnt, start, scan_tmp, &total);
    if (t < (1u << FINE_BITS)) {
      start[t] += beg;
      offsets[(bin << FINE_BITS) + t] = start[t];
    }
    if (t == 0 && bin == COARSE - 1) offsets[MSM_NB] = beg + total;
    __syncthreads();
  }
  // With 256 fine buckets a bucket's run is ~24 entries (96 B): scattered straight to HBM those are partial cache lines from
  // 512 lanes at random times.  A bin that fits the register cache (the usual case) is regrouped in LDS and written out
  // linearly instead; a larger one scatters directly.
  __shared__ uint32_t sorted[(MSM_NB_BITS > 15 && FINE_CACHE > 0) ? FINE_T * FINE_CACHE : 1];
  const bool staged = MSM_NB_BITS > 15 && FINE_CACHE > 0 && end - beg <= FINE_T * FINE_CACHE;
  auto place = [&](WordT e) {
    const uint32_t f = SortWord<WordT>::fine(e);
    const uint32_t pos = start[f] + atomicAdd(&cur[f], 1u);
    if (staged) sorted[pos - beg] = SortWord<WordT>::entry(e);
    else entries[pos] = SortWord<WordT>::entry(e);   // accumulate's format: index | sign << 31
  };
#pragma unroll
  for (uint32_t r = 0; r < FINE_CACHE; ++r)
    if (beg + t + r * FINE_T < end) place(cache[r]);
  for (uint32_t j = beg + t + FINE_CACHE * FINE_T; j < end; j += FINE_T) place(tmp.ld(j));
  if (staged) {
    __syncthreads();
    for (uint32_t j = t; j < end - beg; j += FINE_T) entries[beg + j] = sorted[j];
  }
}

// ---- level 2 for oversized bins: one workgroup per BIG_CHUNK words --------------------------------
// work item w -> (bin, chunk) through the chunk prefix big_off; returns false when w is beyond the last item
__device__ __forceinline__ bool big_item(const uint32_t* __restrict__ big, const uint32_t* __restrict__ coff, uint32_t w,
                                         uint32_t* bin, uint32_t* beg, uint32_t* end) {
  if (w >= big[COARSE]) return false;
  uint32_t lo = 0, hi = COARSE - 1;          // last bin with big[bin] <= w (bins without chunks share their successor's prefix)
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (big[mid] <= w) lo = mid; else hi = mid - 1;
  }
  *bin = lo;
  const uint32_t b0 = coff[lo] + (w - big[lo]) * BIG_CHUNK, b1 = coff[lo + 1];
  *beg = b0;
  *end = b0 + BIG_CHUNK < b1 ? b0 + BIG_CHUNK : b1;
  return true;
}
static constexpr uint32_t BIG_T = 512;
static constexpr uint32_t BIG_PER = BIG_CHUNK / BIG_T;   // words per lane, held in registers
template <class WordT>
__global__ void __launch_bounds__(BIG_T) msm_big_hist_kernel(MsmBatch bt, const uint32_t* __restrict__ coarse_off_all,
                                                             const uint32_t* __restrict__ big_off_all,
                                                             void* __restrict__ tmp_all, uint32_t* __restrict__ big_cnt_all) {
  __shared__ uint32_t cnt[1u << FINE_BITS];
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint32_t* __restrict__ coff = coarse_off_all + (uint64_t)kb * (COARSE + 1);
  const uint32_t* __restrict__ big = big_off_all + (uint64_t)kb * (COARSE + 1);
  const TmpPlanes<WordT> tmp(tmp_all, kb, (uint64_t)MSM_W * bt.cap_m);
  const uint32_t t = threadIdx.x;
  // a SMALL grid walks the chunk list (r04): the list is empty unless the digits are skewed, and one workgroup per POSSIBLE
  // chunk (1 281 x 4 at 2^20 terms) cost 30-45 us of dispatch per launch for nothing
  for (uint32_t item = blockIdx.x;; item += gridDim.x) {
    uint32_t bin, beg, end;
    if (!big_item(big, coff, item, &bin, &beg, &end)) return;   // uniform per workgroup
    if (t < (1u << FINE_BITS)) cnt[t] = 0;
    __syncthreads();
    for (uint32_t j = beg + t; j < end; j += BIG_T) atomicAdd(&cnt[tmp.ld_fine(j)], 1u);
    __syncthreads();
    if (t < (1u << FINE_BITS) && cnt[t]) atomicAdd(&big_cnt_all[(uint64_t)kb * MSM_NB + (bin << FINE_BITS) + t], cnt[t]);
    __syncthreads();
  }
}
template <class WordT>
__global__ void __launch_bounds__(BIG_T) msm_big_scatter_kernel(MsmBatch bt, const uint32_t* __restrict__ coarse_off_all,
                                                                const uint32_t* __restrict__ big_off_all,
                                                                void* __restrict__ tmp_all,
                                                                const uint32_t* __restrict__ big_cnt_all, uint32_t* __restrict__ big_cur_all,
                                                                uint32_t* __restrict__ entries_all, uint32_t* __restrict__ offsets_all) {
  __shared__ uint32_t cnt[1u << FINE_BITS], base[1u << FINE_BITS], cur[1u << FINE_BITS], bin_cnt[1u << FINE_BITS], scan_tmp[1u << FINE_BITS];
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint32_t* __restrict__ coff = coarse_off_all + (uint64_t)kb * (COARSE + 1);
  const uint32_t* __restrict__ big = big_off_all + (uint64_t)kb * (COARSE + 1);
  const TmpPlanes<WordT> tmp(tmp_all, kb, (uint64_t)MSM_W * bt.cap_m);
  uint32_t* __restrict__ entries = entries_all + (uint64_t)kb * MSM_W * bt.cap_m;
  uint32_t* __restrict__ offsets = offsets_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t t = threadIdx.x;
  for (uint32_t item = blockIdx.x;; item += gridDim.x) {   // small grid over the chunk list, see msm_big_hist_kernel
    uint32_t bin, beg, end;
    if (!big_item(big, coff, item, &bin, &beg, &end)) return;
    const uint32_t* __restrict__ bcnt = big_cnt_all + (uint64_t)kb * MSM_NB + (bin << FINE_BITS);
    uint32_t* __restrict__ bcur = big_cur_all + (uint64_t)kb * MSM_NB + (bin << FINE_BITS);
    if (t < (1u << FINE_BITS)) { cnt[t] = 0; cur[t] = 0; }
    __syncthreads();
    WordT cache[BIG_PER];
#pragma unroll
    for (uint32_t r = 0; r < BIG_PER; ++r) {
      const uint32_t j = beg + t + r * BIG_T;
      cache[r] = j < end ? tmp.ld(j) : (WordT)0;
      if (j < end) atomicAdd(&cnt[SortWord<WordT>::fine(cache[r])], 1u);
    }
    __syncthreads();
    {   // bucket starts inside the bin from the bin-wide counts; this chunk's run in every bucket by one atomic each
      const bool first = beg == coff[bin];
      if (t < (1u << FINE_BITS)) bin_cnt[t] = bcnt[t];
      __syncthreads();
      uint32_t total;
      fine_exclusive_scan(bin_cnt, base, scan_tmp, &total);
      if (t < (1u << FINE_BITS)) {
        const uint32_t run = coff[bin] + base[t];
        if (first) offsets[(bin << FINE_BITS) + t] = run;
        base[t] = run + (cnt[t] ? atomicAdd(&bcur[t], cnt[t]) : 0u);
      }
      if (first && t == 0 && bin == COARSE - 1) offsets[MSM_NB] = coff[bin] + total;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < BIG_PER; ++r) {
      if (beg + t + r * BIG_T < end) {
        const WordT e = cache[r];
        const uint32_t f = SortWord<WordT>::fine(e);
        entries[base[f] + atomicAdd(&cur[f], 1u)] = SortWord<WordT>::entry(e);
      }
    }
    __syncthreads();
  }
}

// ---- slice offsets: slice_off[b] = sum_{b' < b} ceil(count[b'] / ksl), one workgroup per commitment
// Also lists the HEAVY buckets (more than heavy_thresh slices: skewed digits) with their 256-slice segments —
// nheavy[2 kb] buckets, nheavy[2 kb + 1] segments — for the segment workers inside msm_bucket_sum (msm.hip).
static constexpr uint32_t HEAVY_SEG_SLICES = 128;   // = HEAVY_SEG of msm.hip
// ORDER: also the layout of msm_accumulate_ordered_kernel — lanes in order of slice LENGTH.  In bucket
// order every wave of msm_accumulate holds a few partial slices (the last slice of each bucket) next to full ones and waits
// for the full ones: 3 % idle lane-steps at 32-entry slices, 6 % at 64, 45 % at m = 2^16 with 32 (tools/slice_order_sim.py).
// Ordered, the full slices come first — full_off = exclusive scan of floor(count / ksl) — and then the partial slice of
// every bucket in order of decreasing length (a counting sort over the ksl - 1 lengths into part_list; part_list[NB] = their
// number).  The partial sums keep their slots slice_off[b] + q, so nothing after the accumulation changes.
// tests/msm_wide_model.py::slice_order is the executable statement of this map.
// Until round 6 a kernel of its own (msm_order_kernel: three passes of dependent reads over the offsets, 51-54 us per
// commitment group after the 24 us of this one); since the rule orders the lanes from 16-entry slices on, every group of a
// 2^17 ... 2^18-gate proof and of a rank of 8 at 2^20 paid it.  Here it shares the batch of offset loads.
// Memory access (round 6, second session): a thread owns PER = NB / 1024 CONSECUTIVE buckets — the scan needs that — so its
// direct loads and stores had a stride of PER words between lanes: every one of the ~130 load / store instructions of a wave
// touched 64 different cache lines (24 us per commitment group for the unordered layout, 38 with the ordering).  The offsets
// are now loaded and the results stored COALESCED through a padded LDS staging array (word i at i + i / 32: the threads'
// strided LDS accesses are conflict-free).
static constexpr uint32_t SLICES_STAGE_WORDS = MSM_NB + MSM_NB / 32 + 2;
static constexpr size_t SLICES_LDS = sizeof(uint32_t) * SLICES_STAGE_WORDS;
__device__ __forceinline__ uint32_t stage_at(uint32_t i) { return i + (i >> 5); }
template <bool ORDER>
__global__ void __launch_bounds__(SORT_T) msm_slices_kernel(const uint32_t* __restrict__ offsets_all,
                                                            uint32_t* __restrict__ slice_off_all, uint32_t ksl, uint32_t heavy_thresh,
                                                            uint32_t* __restrict__ nheavy_all, HeavyItem* __restrict__ heavy_list_all, int kb0,
                                                            uint32_t* __restrict__ full_off_all, uint32_t* __restrict__ part_list_all) {
  extern __shared__ __attribute__((aligned(16))) uint32_t stage[];   // SLICES_STAGE_WORDS
  __shared__ uint32_t sh[SORT_T];
  __shared__ uint32_t hist[129];   // ksl <= 128
  const uint32_t* __restrict__ offsets = offsets_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * (MSM_NB + 1);
  uint32_t* __restrict__ slice_off = slice_off_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * (MSM_NB + 1);
  uint32_t* __restrict__ nheavy = nheavy_all + 2 * (blockIdx.x + (uint32_t)kb0);
  HeavyItem* __restrict__ heavy_list = heavy_list_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * MSM_NB;
  constexpr uint32_t PER = MSM_NB / SORT_T;
  const uint32_t t = threadIdx.x;
  for (uint32_t i = t; i <= MSM_NB; i += SORT_T) stage[stage_at(i)] = offsets[i];
  if (ORDER && t < 129) hist[t] = 0;
  __syncthreads();
  uint32_t o[PER + 1];
#pragma unroll
  for (uint32_t k = 0; k <= PER; ++k) o[k] = stage[stage_at(t * PER + k)];
  const uint32_t sh_ksl = 31u - (uint32_t)__builtin_clz(ksl);   // ksl is a power of two (msm_ksl)
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) mine += (o[k + 1] - o[k] + ksl - 1) >> sh_ksl;
  uint32_t total;
  uint32_t run = block_exclusive_scan(mine, sh, &total);   // (its barriers: every thread has read its offsets from the staging array)
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) {
    stage[stage_at(t * PER + k)] = run;
    const uint32_t ns = (o[k + 1] - o[k] + ksl - 1) >> sh_ksl;
    run += ns;
    if (ns > heavy_thresh) {
      HeavyItem it;
      it.bucket = t * PER + k;
      it.nseg = (ns + HEAVY_SEG_SLICES - 1) / HEAVY_SEG_SLICES;
      it.seg_base = atomicAdd(&nheavy[1], it.nseg);
      it.pad = 0;
      heavy_list[atomicAdd(&nheavy[0], 1u)] = it;
    }
  }
  __syncthreads();
  for (uint32_t i = t; i < MSM_NB; i += SORT_T) slice_off[i] = stage[stage_at(i)];
  if (t == SORT_T - 1) slice_off[MSM_NB] = total;
  if constexpr (ORDER) {
    uint32_t* __restrict__ full_off = full_off_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * (MSM_NB + 1);
    uint32_t* __restrict__ part_list = part_list_all + (uint64_t)(blockIdx.x + (uint32_t)kb0) * (MSM_NB + 1);
    uint32_t mine_f = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t c = o[k + 1] - o[k];
      mine_f += c >> sh_ksl;
      const uint32_t r = c & (ksl - 1);
      if (r) atomicAdd(&hist[r], 1u);
    }
    uint32_t total_f;
    uint32_t run_f = block_exclusive_scan(mine_f, sh, &total_f);   // (its barriers also complete the histogram and the copy-out above)
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      stage[stage_at(t * PER + k)] = run_f;
      run_f += (o[k + 1] - o[k]) >> sh_ksl;
    }
    if (t == 0) {   // start of every length class, longest first; hist[0] = number of partial slices
      uint32_t acc = 0;
      for (uint32_t r = ksl - 1; r >= 1; --r) { const uint32_t h = hist[r]; hist[r] = acc; acc += h; }
      hist[0] = acc;
    }
    __syncthreads();
    for (uint32_t i = t; i < MSM_NB; i += SORT_T) full_off[i] = stage[stage_at(i)];
    if (t == SORT_T - 1) full_off[MSM_NB] = total_f;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
      const uint32_t r = (o[k + 1] - o[k]) & (ksl - 1);
      if (r) part_list[atomicAdd(&hist[r], 1u)] = t * PER + k;
    }
    if (t == 0) part_list[MSM_NB] = hist[0];
  }
}
#if PLONK_MSM_NB_BITS > 15
// ---- the same two layouts (slice offsets + heavy list, full-slice offsets + partial slices by length) for MANY buckets --
// The two single-workgroup kernels above walk the bucket counts with 1024 threads x NB / 1024 strided reads each: fine for
// 2^15 buckets (40-70 us), a millisecond each for 2^19 (r03c first cut: 1.09 + 0.97 ms per launch).  Here: one workgroup
// per 1024 buckets counts and scans locally (coalesced), one workgroup per commitment scans the 512 block totals and the
// 512 x (ksl - 1) remainder-class counts, and a third pass applies the bases.  layout: [NBLK] slices | [NBLK] full |
// [NBLK][LAY_H] class counts -> exclusive prefixes | [LAY_H] class starts.
static constexpr uint32_t LAY_NBLK = MSM_NB / SORT_T, LAY_H = 132;
static constexpr uint32_t LAY_WORDS = 2 * LAY_NBLK + (LAY_NBLK + 1) * LAY_H;
__global__ void __launch_bounds__(SORT_T) msm_layout_count_kernel(const uint32_t* __restrict__ offsets_all, uint32_t* __restrict__ slice_off_all,
                                                                  uint32_t* __restrict__ full_off_all, uint32_t* __restrict__ lay_all, uint32_t ksl, int kb0) {
  __shared__ uint32_t sh[SORT_T];
  __shared__ uint32_t hist[LAY_H];
  const int kb = (int)blockIdx.y + kb0;
  const uint32_t blk = blockIdx.x, t = threadIdx.x, b = blk * SORT_T + t;
  const uint32_t* __restrict__ offsets = offsets_all + (uint64_t)kb * (MSM_NB + 1);
  uint32_t* __restrict__ lay = lay_all + (uint64_t)kb * LAY_WORDS;
  if (t < LAY_H) hist[t] = 0;
  __syncthreads();
  const uint32_t cnt = offsets[b + 1] - offsets[b];
  const uint32_t nf = cnt / ksl, r = cnt % ksl, ns = nf + (r ? 1u : 0u);
  if (r) atomicAdd(&hist[r], 1u);
  uint32_t tot_ns, tot_nf;
  const uint32_t ex_ns = block_exclusive_scan(ns, sh, &tot_ns);
  const uint32_t ex_nf = block_exclusive_scan(nf, sh, &tot_nf);
  slice_off_all[(uint64_t)kb * (MSM_NB + 1) + b] = ex_ns;   // local to the block until msm_layout_apply adds the base
  full_off_all[(uint64_t)kb * (MSM_NB + 1) + b] = ex_nf;
  if (t == 0) { lay[blk] = tot_ns; lay[LAY_NBLK + blk] = tot_nf; }
  if (t < LAY_H) lay[2 * LAY_NBLK + blk * LAY_H + t] = hist[t];   // (the scans' barriers completed the histogram)
}
__global__ void __launch_bounds__(SORT_T) msm_layout_scan_kernel(uint32_t* __restrict__ slice_off_all, uint32_t* __restrict__ full_off_all,
                                                                 uint32_t* __restrict__ part_list_all, uint32_t* __restrict__ lay_all, uint32_t ksl, int kb0) {
  __shared__ uint32_t sh[SORT_T];
  __shared__ uint32_t ctot[LAY_H];
  const int kb = (blockIdx.x + (uint32_t)kb0);
  const uint32_t t = threadIdx.x;
  uint32_t* __restrict__ lay = lay_all + (uint64_t)kb * LAY_WORDS;
  static_assert(LAY_NBLK <= SORT_T, "one thread per block total");
  uint32_t tot;
  const uint32_t a = t < LAY_NBLK ? lay[t] : 0u;
  const uint32_t ea = block_exclusive_scan(a, sh, &tot);
  if (t < LAY_NBLK) lay[t] = ea;
  if (t == 0) slice_off_all[(uint64_t)kb * (MSM_NB + 1) + MSM_NB] = tot;
  const uint32_t f = t < LAY_NBLK ? lay[LAY_NBLK + t] : 0u;
  const uint32_t ef = block_exclusive_scan(f, sh, &tot);
  if (t < LAY_NBLK) lay[LAY_NBLK + t] = ef;
  if (t == 0) full_off_all[(uint64_t)kb * (MSM_NB + 1) + MSM_NB] = tot;
  if (t < LAY_H) {   // remainder class t: exclusive prefix over the blocks (reads coalesced across the classes)
    uint32_t run = 0;
    if (t >= 1 && t < ksl)
      for (uint32_t blk = 0; blk < LAY_NBLK; ++blk) {
        uint32_t* q = lay + 2 * LAY_NBLK + blk * LAY_H + t;
        const uint32_t h = *q;
        *q = run;
        run += h;
      }
    ctot[t] = run;
  }
  __syncthreads();
  if (t == 0) {   // start of every class, longest first
    uint32_t acc = 0;
    uint32_t* cstart = lay + 2 * LAY_NBLK + LAY_NBLK * LAY_H;
    for (uint32_t r = ksl - 1; r >= 1; --r) { cstart[r] = acc; acc += ctot[r]; }
    part_list_all[(uint64_t)kb * (MSM_NB + 1) + MSM_NB] = acc;
  }
}
__global__ void __launch_bounds__(SORT_T) msm_layout_apply_kernel(const uint32_t* __restrict__ offsets_all, uint32_t* __restrict__ slice_off_all,
                                                                  uint32_t* __restrict__ full_off_all, uint32_t* __restrict__ part_list_all,
                                                                  const uint32_t* __restrict__ lay_all, uint32_t ksl, uint32_t heavy_thresh,
                                                                  uint32_t* __restrict__ nheavy_all, HeavyItem* __restrict__ heavy_list_all,
                                                                  uint4* __restrict__ buckets_raw, uint32_t* __restrict__ multi_list_all, int kb0) {
  __shared__ uint32_t rank[LAY_H];
  const int kb = (int)blockIdx.y + kb0;
  const uint32_t blk = blockIdx.x, t = threadIdx.x, b = blk * SORT_T + t;
  const uint32_t* __restrict__ offsets = offsets_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t* __restrict__ lay = lay_all + (uint64_t)kb * LAY_WORDS;
  if (t < LAY_H) rank[t] = 0;
  __syncthreads();
  const uint32_t cnt = offsets[b + 1] - offsets[b];
  const uint32_t nf = cnt / ksl, r = cnt % ksl, ns = nf + (r ? 1u : 0u);
  slice_off_all[(uint64_t)kb * (MSM_NB + 1) + b] += lay[blk];
  full_off_all[(uint64_t)kb * (MSM_NB + 1) + b] += lay[LAY_NBLK + blk];
  if (r) {
    const uint32_t pos = lay[2 * LAY_NBLK + LAY_NBLK * LAY_H + r] + lay[2 * LAY_NBLK + blk * LAY_H + r] + atomicAdd(&rank[r], 1u);
    part_list_all[(uint64_t)kb * (MSM_NB + 1) + pos] = b;
  }
  if (ns > heavy_thresh) {
    HeavyItem it;
    it.bucket = b;
    it.nseg = (ns + HEAVY_SEG_SLICES - 1) / HEAVY_SEG_SLICES;
    it.seg_base = atomicAdd(&nheavy_all[2 * kb + 1], it.nseg);
    it.pad = 0;
    heavy_list_all[(uint64_t)kb * MSM_NB + atomicAdd(&nheavy_all[2 * kb], 1u)] = it;
  } else if (ns == 0) {
    uint4* q = buckets_raw + ((uint64_t)kb * MSM_NB + b) * 16;   // empty bucket: the identity (ZZ = 0) — a 256-B slot of zeros
#pragma unroll
    for (int k = 0; k < 16; ++k) q[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  // with ~24 entries per bucket a bucket is ONE slice and its lane of msm_accumulate_ordered writes the bucket itself; the
  // few with a second slice are listed so that msm_bucket_sum touches only them (nheavy[2 KB + kb] = their number).  One
  // global atomic per workgroup: ~26 k single appends on one counter cost 0.2 ms per launch (r03c).
  __shared__ uint32_t nmul, mbase;
  if (t == 0) nmul = 0;
  __syncthreads();
  const bool multi = ns >= 2 && ns <= heavy_thresh;
  const uint32_t mrank = multi ? atomicAdd(&nmul, 1u) : 0u;
  __syncthreads();
  if (t == 0 && nmul) mbase = atomicAdd(&nheavy_all[2 * MSM_MAX_BATCH + kb], nmul);
  __syncthreads();
  if (multi) multi_list_all[(uint64_t)kb * MSM_NB + mbase + mrank] = b;
}
#endif

int msm_order_slices(Ctx* c, const MsmBatch& bt) {
  (void)c; (void)bt;
  return PLONK_OK;   // msm_group_sort already wrote full_off / part_list (bt.ordered): the layout kernels (many buckets) or msm_slices_kernel<true>
}

// Host side: everything between the scalars and msm_accumulate for one commitment group.
template <int MODE, class WordT>
static int msm_group_sort_t(Ctx* c, const MsmBatch& bt, uint64_t mmax) {
  MsmWork& w = c->msm;
  hipStream_t st = c->stream;
  const uint32_t tiles = (uint32_t)((mmax + TILE - 1) / TILE);
  const uint32_t htiles = (uint32_t)((mmax + HIST_TILE - 1) / HIST_TILE);
  void* tmp = (void*)w.tmp_words;
  // bt.kb0 > 0 (a group launched column by column, msm_batch_device's phases): this launch covers commitments
  // [kb0, kb0 + count) of the group's buffers — every kernel adds kb0 to its blockIdx-derived commitment index
  const int kb0 = bt.kb0;
  // (Round 6, measured and NOT adopted: clearing the three counter arrays inside msm_coarse_scan_kernel instead of by three
  // hipMemsetAsync launches — 2^12: 2.577 -> 2.556 ms, 2^16: 4.778 -> 4.759, a rank of 8 at 2^20: 6.77 -> 6.75, i.e. < 1 %
  // for a zero-between-groups invariant shared by both bucket-count variants; profiles/r06/tidy_ab.jsonl.)
  HIP_TRY(hipMemsetAsync(w.coarse_cnt + (size_t)COARSE * kb0, 0, sizeof(uint32_t) * COARSE * bt.count, st));
  hipLaunchKernelGGL(msm_hist_kernel<MODE>, dim3(htiles, bt.count), dim3(SORT_T), 0, st, bt, w.coarse_cnt);
  hipLaunchKernelGGL(msm_coarse_scan_kernel, dim3(bt.count), dim3(SORT_T), 0, st, w.coarse_cnt, w.coarse_off, w.coarse_cur, w.big_off, kb0);
#if PLONK_MSM_NB_BITS >= 19
  if constexpr (MODE != 0) {
    if (c->cfg.sort13 == 1) {   // round 5 A/B: two half-size partition workgroups per CU (13 digit slots of 1024 scalars)
      const uint32_t tiles2 = (uint32_t)((mmax + P2_TILE - 1) / P2_TILE);
      smem_opt_in(c, (const void*)msm_partition2_kernel<MODE, WordT>, PARTITION2_LDS);
      hipLaunchKernelGGL((msm_partition2_kernel<MODE, WordT>), dim3(tiles2, bt.count), dim3(P2_T), PARTITION2_LDS, st, bt, bt.table_n,
                         w.coarse_off, w.coarse_cur, tmp);
    }
  }
  if (MODE == 0 || c->cfg.sort13 != 1) {
#else
  {
#endif
    smem_opt_in(c, (const void*)msm_partition_kernel<MODE, WordT>, PARTITION_LDS);
    hipLaunchKernelGGL((msm_partition_kernel<MODE, WordT>), dim3(tiles, bt.count), dim3(SORT_T), PARTITION_LDS, st, bt, bt.table_n,
                       w.coarse_off, w.coarse_cur, tmp);
  }
  hipLaunchKernelGGL(msm_fine_kernel<WordT>, dim3(COARSE, bt.count), dim3(FINE_T), 0, st, bt, w.coarse_off, tmp,
                     w.entries, w.offsets);
  {   // oversized bins (skewed digits): upper bound of the chunk count known on the host, surplus workgroups exit at once
    const uint64_t words = (uint64_t)MSM_W * mmax;
    const uint32_t most = (uint32_t)(words / BIG_CHUNK + words / BIG_LIMIT + 1);
    const uint32_t big_wgs = most < 512u ? most : 512u;   // the kernels stride over the chunk list
    if (kb0 == 0 && bt.count == bt.group_count) {
      HIP_TRY(hipMemsetAsync(w.big_cnt, 0, sizeof(uint32_t) * 2 * MSM_NB * MSM_MAX_BATCH, st));   // big_cnt | big_cur
    } else {   // a column launch clears only its own counters (the other columns' may be in use on another stream one day)
      HIP_TRY(hipMemsetAsync(w.big_cnt + (size_t)MSM_NB * kb0, 0, sizeof(uint32_t) * MSM_NB * bt.count, st));
      HIP_TRY(hipMemsetAsync(w.big_cnt + (size_t)MSM_NB * MSM_MAX_BATCH + (size_t)MSM_NB * kb0, 0, sizeof(uint32_t) * MSM_NB * bt.count, st));
    }
    hipLaunchKernelGGL(msm_big_hist_kernel<WordT>, dim3(big_wgs, bt.count), dim3(BIG_T), 0, st, bt, w.coarse_off, w.big_off, tmp, w.big_cnt);
    hipLaunchKernelGGL(msm_big_scatter_kernel<WordT>, dim3(big_wgs, bt.count), dim3(BIG_T), 0, st, bt, w.coarse_off, w.big_off, tmp,
                       w.big_cnt, w.big_cnt + (size_t)MSM_NB * MSM_MAX_BATCH, w.entries, w.offsets);
  }
  if (kb0 == 0 && bt.count == bt.group_count) {
    HIP_TRY(hipMemsetAsync(w.nheavy, 0, sizeof(uint32_t) * 4 * MSM_MAX_BATCH, st));
  } else {   // nheavy[2 kb], nheavy[2 kb + 1] (heavy buckets / segments) and nheavy[2 KB + kb] (multi-slice buckets) of these columns
    HIP_TRY(hipMemsetAsync(w.nheavy + 2 * kb0, 0, sizeof(uint32_t) * 2 * bt.count, st));
    HIP_TRY(hipMemsetAsync(w.nheavy + 2 * MSM_MAX_BATCH + kb0, 0, sizeof(uint32_t) * bt.count, st));
  }
#if PLONK_MSM_NB_BITS > 15
  if (bt.ksl > 128) return (set_last_error("msm_group_sort", "slice length above 128", __FILE__, __LINE__), PLONK_ERR_ARG);
  hipLaunchKernelGGL(msm_layout_count_kernel, dim3(LAY_NBLK, bt.count), dim3(SORT_T), 0, st, w.offsets, w.slice_off, w.full_off, w.layout, bt.ksl, kb0);
  hipLaunchKernelGGL(msm_layout_scan_kernel, dim3(bt.count), dim3(SORT_T), 0, st, w.slice_off, w.full_off, w.part_list, w.layout, bt.ksl, kb0);
  hipLaunchKernelGGL(msm_layout_apply_kernel, dim3(LAY_NBLK, bt.count), dim3(SORT_T), 0, st, w.offsets, w.slice_off, w.full_off, w.part_list,
                     (const uint32_t*)w.layout, bt.ksl, bt.heavy_thresh, w.nheavy, (HeavyItem*)w.heavy_list, (uint4*)w.buckets, w.multi_list, kb0);
#else
  if (bt.ordered) {
    if (bt.ksl > 128) return (set_last_error("msm_group_sort", "slice length above 128", __FILE__, __LINE__), PLONK_ERR_ARG);
    smem_opt_in(c, (const void*)msm_slices_kernel<true>, SLICES_LDS);
    hipLaunchKernelGGL(msm_slices_kernel<true>, dim3(bt.count), dim3(SORT_T), SLICES_LDS, st, w.offsets, w.slice_off, bt.ksl, bt.heavy_thresh,
                       w.nheavy, (HeavyItem*)w.heavy_list, kb0, w.full_off, w.part_list);
  } else {
    smem_opt_in(c, (const void*)msm_slices_kernel<false>, SLICES_LDS);
    hipLaunchKernelGGL(msm_slices_kernel<false>, dim3(bt.count), dim3(SORT_T), SLICES_LDS, st, w.offsets, w.slice_off, bt.ksl, bt.heavy_thresh,
                       w.nheavy, (HeavyItem*)w.heavy_list, kb0, (uint32_t*)nullptr, (uint32_t*)nullptr);
  }
#endif
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
int msm_group_sort(Ctx* c, const MsmBatch& bt, uint64_t mmax) {
  // final entries: sign << 31 | 31-bit table index
  if ((uint64_t)bt.rows * bt.table_n > (1ull << 31))
    return (set_last_error("commit key too large for the bucket sort", "table rows * points must be <= 2^31", __FILE__, __LINE__), PLONK_ERR_ARG);
  if (bt.rows == MSM_ROWS_BITPOS) return bt.wide ? msm_group_sort_t<1, uint64_t>(c, bt, mmax) : msm_group_sort_t<1, uint32_t>(c, bt, mmax);
  if (bt.rows == MSM_ROWS_HALFPOS) return bt.wide ? msm_group_sort_t<2, uint64_t>(c, bt, mmax) : msm_group_sort_t<2, uint32_t>(c, bt, mmax);
#if PLONK_MSM_NB_BITS == 15
  return bt.wide ? msm_group_sort_t<0, uint64_t>(c, bt, mmax) : msm_group_sort_t<0, uint32_t>(c, bt, mmax);
#else
  return (set_last_error("msm_group_sort", "window tables need the 2^15-bucket variant", __FILE__, __LINE__), PLONK_ERR_ARG);
#endif
}
// the narrow (32-bit) intermediate word holds IDX_BITS bits of table index
bool msm_needs_wide_words(uint32_t rows, uint64_t table_n) { return (uint64_t)rows * table_n > (1ull << IDX_BITS); }

}  // namespace PLONK_MSM_NS
}  // namespace plonk
