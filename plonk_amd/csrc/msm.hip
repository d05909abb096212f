// KZG10 commitment MSM on G1 for gfx950 (MI355X): sum_i s_i * [tau^i]G.
//
// Replaces dusk_bls12_381::multiscalar_mul::msm_variable_base as called from
// CommitKey::commit (reference src/commitment_scheme/kzg10/key.rs:376-388).
// The result is a unique group element, so the schedule is free; this one is
// sized for 288 GB of HBM rather than copied from the CPU Pippenger:
//
//   * plonk_srs_load precomputes one table row per BIT POSITION, T[r][i] = 2^r * P_i, r < 256 (128 B per entry:
//     32 GiB at 2^20 points), when that fits comfortably in the free HBM, else the 16 window rows
//     T[w][i] = 2^(16 w) * P_i (2 GiB).  All rows share ONE set of 2^15 signed-digit buckets, so there is no
//     per-window bucket reduction and no Horner doubling chain at the end; with a row per bit the digits are a
//     width-17 NAF — 14.7 instead of 16 additions per scalar (msm_recode.cuh).
//   * bucket grouping (msm_sort.hip): scalars -> at most 16 signed digits -> table entries
//                    (index | sign) grouped by bucket with a hand-written two-level counting
//                    sort; bucket offsets and slice offsets (a slice = at most MSM_KSL entries
//                    of one bucket) fall out of it.
//   * msm_accumulate (dominant): one lane per slice; gathers affine table points
//                    (128-byte entries, one cache line each) and folds them into an
//                    XYZZ accumulator.  Field arithmetic is the reduced-radix, lazily
//                    reduced Fp28 of fp28.cuh / curve28.cuh: a mixed addition is ten
//                    Montgomery products of 2*14*14 v_mad_u64_u32 each and no carry chains.
//   * msm_bucket_sum: slice partials -> buckets (one full addition per slice: throughput-bound);
//     msm_heavy_seg / msm_heavy_bucket: buckets with far more slices than expected (skewed digits);
//     msm_rowcol -> msm_bits (or msm_final): sum_b b * B_b as row / column sums and 16 bit sums that the host
//     finishes.  These are chains of DEPENDENT additions; the default kernels (*_quad) run every addition on the four
//     lanes of a quad, one product per lane and level (g1r_add_quad), ~2.8x lower latency per addition.
//
// Algorithmic HBM bytes per MSM of m terms: 128 * m (32 B scalar + 96 B base).
#include <cstdlib>

#include "plonk_internal.hpp"
#include "curve28.cuh"
#include "fr29.cuh"

namespace plonk {


static constexpr uint32_t MSM_KSL = 32;   // entries per slice from 2^18 terms on (msm_ksl: shorter slices, 3-6 per bucket, for smaller m)
static constexpr uint32_t MSM_CHUNK = 16; // buckets per chunk in the weighted reduction

__device__ __forceinline__ Fr ld_fr_g(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ Fp ld_fp(const Fp* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1], c = q[2];
  Fp r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
  return r;
}
__device__ __forceinline__ void st_fp(Fp* p, const Fp& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}
__device__ __forceinline__ G1Affine ld_aff(const G1Affine* p) {
  G1Affine a;
  a.x = ld_fp(&p->x);
  a.y = ld_fp(&p->y);
  return a;
}
__device__ __forceinline__ void st_aff(G1Affine* p, const G1Affine& a) {
  st_fp(&p->x, a.x);
  st_fp(&p->y, a.y);
}
__device__ __forceinline__ G1 ld_g1(const G1* p) {
  G1 r;
  r.X = ld_fp(&p->X); r.Y = ld_fp(&p->Y); r.ZZ = ld_fp(&p->ZZ); r.ZZZ = ld_fp(&p->ZZZ);
  return r;
}
__device__ __forceinline__ void st_g1(G1* p, const G1& v) {
  st_fp(&p->X, v.X); st_fp(&p->Y, v.Y); st_fp(&p->ZZ, v.ZZ); st_fp(&p->ZZZ, v.ZZZ);
}

// ---- Fp28 / G1R in memory: each coordinate padded to 16 words (64 B) ------------------
struct alignas(16) G1RSlot {
  Fp28Slot X, Y, ZZ, ZZZ;
};
__device__ __forceinline__ Fp28 ld_f28(const Fp28Slot* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  Fp28 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
  r.l[12] = d.x; r.l[13] = d.y;
  return r;
}
__device__ __forceinline__ void st_f28(Fp28Slot* p, const Fp28& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
  q[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
__device__ __forceinline__ G1R ld_g1r(const G1RSlot* p) {
  G1R r;
  r.X = ld_f28(&p->X); r.Y = ld_f28(&p->Y); r.ZZ = ld_f28(&p->ZZ); r.ZZZ = ld_f28(&p->ZZZ);
  return r;
}
__device__ __forceinline__ void st_g1r(G1RSlot* p, const G1R& v) {
  st_f28(&p->X, v.X); st_f28(&p->Y, v.Y); st_f28(&p->ZZ, v.ZZ); st_f28(&p->ZZZ, v.ZZZ);
}
__device__ __forceinline__ G1R g1r_neg(const G1R& p) {   // -P: Y -> 16p - Y (Y < 8p), canonicalised
  G1R r = p;
  if (!p.is_identity()) r.Y = Fp28::sub<16>(Fp28::zero(), p.Y).canon();
  return r;
}

#if PLONK_MSM_NB_BITS == 15   // ---- shared code: compiled once (the 2^15-bucket build of this file) ----
// ---------------------------------------------------------------------------
// SRS tables
// ---------------------------------------------------------------------------
// T[r * n + i] = 2^(r * step) * P_i, affine, in the reduced-radix form (x, y < 2p): `rows` = 16, step = 16 (window
// tables) or rows = 256, step = 1 (bit-position tables).  One lane per point: (rows - 1) * step doublings in XYZZ
// coordinates, and ONE Fp inversion for the rows - 1 normalisations (Montgomery's trick along the lane's own rows: the
// unnormalised X, Y wait in their table slots, ZZ, ZZZ and the running product of the ZZ * ZZZ in a scratch array of
// 3 x 64 B per row and point; one inversion per row made the inversions 4/5 of this kernel; the one that remains is the
// safegcd inverse of fp_safegcd.cuh).  The doubling chain has the same length for both kinds of table (240 / 255
// doublings); bit-position tables store — and normalise — every step of it.  One-off per commit key, outside every
// timed region.  `pts` holds points [first, first + count) of the key (a chunk of the stream in plonk_srs_load); row r
// starts at r * n.  scratch: [(rows - 1) * 3][count] slots.
static uint64_t srs_table_chunk_points(uint32_t rows) {   // points per launch: bounds the scratch (720 MiB / 3.1 GiB)
  return rows == MSM_ROWS_WINDOW ? 1ull << 18 : 1ull << 16;
}
__global__ void srs_table_kernel(const G1Affine* __restrict__ pts, G1AffineR* __restrict__ table, uint64_t n,
                                 uint64_t first, uint64_t count, Fp28Slot* __restrict__ scratch, uint32_t rows, uint32_t step) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  const uint64_t i = first + j;
  const G1Affine a = ld_aff(pts + j);
  G1R p = G1R::from_affine(Fp28::from_fp(a.x), Fp28::from_fp(a.y));
  st_f28(&table[i].x, p.X);
  st_f28(&table[i].y, p.Y);
  Fp28 run = Fp28::one();
  for (uint32_t w = 1; w < rows; ++w) {
    for (uint32_t k = 0; k < step; ++k) p = p.dbl();          // P_i has prime order: never the identity
    Fp28Slot* sc = scratch + (uint64_t)(w - 1) * 3 * count + j;
    st_f28(&table[(uint64_t)w * n + i].x, p.X.normalized());
    st_f28(&table[(uint64_t)w * n + i].y, p.Y.normalized());
    st_f28(sc, p.ZZ);
    st_f28(sc + count, p.ZZZ);
    run = Fp28::mul(run, Fp28::mul(p.ZZ, p.ZZZ));
    st_f28(sc + 2 * count, run);
  }
  Fp28 inv = fp28_inv_gcd(run);                            // 1 / prod_w ZZ_w ZZZ_w
  for (uint32_t w = rows - 1; w >= 1; --w) {
    const Fp28Slot* sc = scratch + (uint64_t)(w - 1) * 3 * count + j;
    const Fp28 zz = ld_f28(sc), zzz = ld_f28(sc + count);
    const Fp28 before = w > 1 ? ld_f28(sc - 3 * count + 2 * count) : Fp28::one();   // running product up to w - 1
    const Fp28 iw = Fp28::mul(inv, before);                // 1 / (ZZ_w ZZZ_w)
    inv = Fp28::mul(inv, Fp28::mul(zz, zzz));
    G1AffineR* e = &table[(uint64_t)w * n + i];
    st_f28(&e->x, Fp28::mul(ld_f28(&e->x), Fp28::mul(iw, zzz)));   // X / ZZ
    st_f28(&e->y, Fp28::mul(ld_f28(&e->y), Fp28::mul(iw, zz)));    // Y / ZZZ
  }
}

// CommitKey::from_raw_var_bytes (key.rs:263-300) validates every decoded point with
// is_on_curve() & is_torsion_free(); here one lane per point: y^2 = x^3 + 4 and [q]P = O
// (255 doublings + one mixed addition per set bit of q).  flag |= 1 on any failure.
__global__ void srs_validate_kernel(const G1Affine* __restrict__ pts, uint64_t n, int* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const G1Affine a = ld_aff(pts + i);
  const Fp28 x = Fp28::from_fp(a.x), y = Fp28::from_fp(a.y);
  const Fp28 one = Fp28::one();
  const Fp28 four = Fp28::add(Fp28::add(one, one), Fp28::add(one, one));
  const Fp28 rhs = Fp28::add(Fp28::mul(x.sqr(), x), four);          // < 6p
  bool ok = Fp28::sub<16>(y.sqr(), rhs).is_zero_mod();
  if (ok) {
    G1R acc = G1R::from_affine(x, y);                                // top bit of q (bit 254)
    for (int b = 253; b >= 0; --b) {
      acc = acc.dbl();
      if ((FrP::MOD[b >> 5] >> (b & 31)) & 1) acc = acc.add_affine(x, y);
    }
    ok = acc.is_identity();
  }
  if (!ok) atomicOr(flag, 1);
}

__device__ __forceinline__ G1Affine g1_generator() {
  G1Affine g;
  const uint32_t gx[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                           0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
  const uint32_t gy[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                           0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
#pragma unroll
  for (int k = 0; k < 12; ++k) { g.x.l[k] = gx[k]; g.y.l[k] = gy[k]; }
  return g;
}

// out[i] = (g_scalar * tau^i) * G   — synthetic SRS (srs.rs:61-100 semantics)
__global__ void srs_generate_kernel(Fr tau, Fr g_scalar, uint64_t n, G1Affine* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr k = (g_scalar * tau.pow_u64(i)).from_mont();
  const G1Affine g = g1_generator();
  G1 acc = G1::identity();
  for (int w = 7; w >= 0; --w)
    for (int b = 31; b >= 0; --b) {
      acc = acc.dbl();
      if ((k.l[w] >> b) & 1) acc = acc.add_affine(g);
    }
  G1Affine a;
  acc.to_affine(&a);
  st_aff(out + i, a);
}

#endif   // shared

namespace PLONK_MSM_NS {   // ---- per bucket count: accumulation, bucket sums, reduction tail ----
// ---------------------------------------------------------------------------
// accumulation
// ---------------------------------------------------------------------------
// The first TWO entries of a lane are both affine table points: their sum takes 4 products + 2 squarings with 5 reductions
// (G1R::add_affine_pair) instead of the copy + the general mixed addition's 8 + 2 with 9 — four of those products were
// multiplications by ZZ = ZZZ = 1 (round 6, second session).  One addition in ~23 per lane at 2^20 gates, one in 8 at 2^16.
// The macro consumes entries k0 and k0 + 1, leaves the NEXT entry prefetched in (ent, x, y) as the loop expects and advances
// k0; a lane with one entry, or whose first two points share their x (equal or opposite: the general path doubles or
// cancels), is left untouched.
#define ACC_FIRST_PAIR(acc, ent, x, y, k0, end)                                                                   \
  do {                                                                                                            \
    if ((k0) + 1 < (end)) {                                                                                       \
      const uint32_t ent_b_ = entries[(k0) + 1];                                                                  \
      const Fp28 xb_ = ld_f28(&table[ent_b_ & 0x7fffffffu].x);                                                    \
      const Fp28 yb_ = ld_f28(&table[ent_b_ & 0x7fffffffu].y);                                                    \
      if (G1R::pair_distinct((x), xb_)) {                                                                         \
        const uint32_t ent_a_ = (ent);                                                                            \
        const Fp28 xa_ = (x), ya_ = (y);                                                                          \
        if ((k0) + 2 < (end)) {                                                                                   \
          (ent) = entries[(k0) + 2];                                                                              \
          (x) = ld_f28(&table[(ent) & 0x7fffffffu].x);                                                            \
          (y) = ld_f28(&table[(ent) & 0x7fffffffu].y);                                                            \
        }                                                                                                         \
        (acc) = G1R::add_affine_pair(xa_, acc_signed_y(ya_, (ent_a_ & 0x80000000u) != 0), xb_,                    \
                                     acc_signed_y(yb_, (ent_b_ & 0x80000000u) != 0));                             \
        (k0) += 2;                                                                                                \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
__device__ __forceinline__ Fp28 acc_signed_y(const Fp28& y, bool neg);
__global__ void __launch_bounds__(128) msm_accumulate_kernel(const G1AffineR* __restrict__ table, MsmBatch bt,
                                                             const uint32_t* __restrict__ entries_all,
                                                             const uint32_t* __restrict__ offsets_all,
                                                             const uint32_t* __restrict__ slice_off_all,
                                                             G1RSlot* __restrict__ partial_all) {
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint32_t* __restrict__ entries = entries_all + (uint64_t)kb * MSM_W * bt.cap_m;
  const uint32_t* __restrict__ offsets = offsets_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t* __restrict__ slice_off = slice_off_all + (uint64_t)kb * (MSM_NB + 1);
  G1RSlot* __restrict__ partial = partial_all + (uint64_t)kb * bt.cap_slices;
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nslices = slice_off[MSM_NB];
  // bucket of this slice: largest b with slice_off[b] <= s.  The 128 slices of a workgroup are
  // consecutive, so their buckets lie in one narrow range: every 64th offset is staged in LDS
  // (9 of the 15 search steps never leave the CU), the last 6 steps read global memory.
  __shared__ uint32_t coarse[MSM_NB / 64];
  for (uint32_t j = threadIdx.x; j < MSM_NB / 64; j += blockDim.x) coarse[j] = slice_off[j * 64];
  __syncthreads();
  if (s >= nslices) return;
  uint32_t lo = 0, hi = MSM_NB / 64 - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (coarse[mid] <= s) lo = mid; else hi = mid - 1;
  }
  lo *= 64;
  hi = lo + 63;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (slice_off[mid] <= s) lo = mid; else hi = mid - 1;
  }
  const uint32_t b = lo;
  const uint32_t q = s - slice_off[b];
  const uint32_t beg = offsets[b] + q * bt.ksl;
  uint32_t end = beg + bt.ksl;
  const uint32_t bend = offsets[b + 1];
  if (end > bend) end = bend;
  G1R acc = G1R::identity();
  // software pipeline: the table entry of step k+1 is in flight while step k's ~5k VALU
  // instructions run (two waves per SIMD are not enough to hide a 128-B gather otherwise)
  uint32_t ent = entries[beg];
  Fp28 x = ld_f28(&table[ent & 0x7fffffffu].x);
  Fp28 y = ld_f28(&table[ent & 0x7fffffffu].y);
  uint32_t k0 = beg;
  ACC_FIRST_PAIR(acc, ent, x, y, k0, end);
  for (uint32_t k = k0; k < end; ++k) {
    const uint32_t ent_c = ent;
    const Fp28 xc = x, yc = y;
    if (k + 1 < end) {
      ent = entries[k + 1];
      x = ld_f28(&table[ent & 0x7fffffffu].x);
      y = ld_f28(&table[ent & 0x7fffffffu].y);
    }
    // -y as 4p - y without carry propagation (lazy limbs): it only feeds a product
    Fp28 y2;
    const bool neg = ent_c & 0x80000000u;
#pragma unroll
    for (int i = 0; i < Fp28::N; ++i) y2.l[i] = neg ? Fp28::pad<4>(i) - yc.l[i] : yc.l[i];
    acc = acc.add_affine(xc, y2);
  }
  st_g1r(partial + s, acc);
}

// msm_accumulate_kernel with the lanes in order of slice length (PLONK_MSM_ORDER=1; msm_sort.hip msm_slices_kernel<true>): lane
// s < F owns full slice s - full_off[b] of the bucket b found by the same two-level search, the lanes after them own the
// partial slice of bucket part_list[s - F].  The additions are the ones above; the partial sum goes to the slice's usual
// slot slice_off[b] + q.
__global__ void __launch_bounds__(128) msm_accumulate_ordered_kernel(const G1AffineR* __restrict__ table, MsmBatch bt,
                                                                     const uint32_t* __restrict__ entries_all,
                                                                     const uint32_t* __restrict__ offsets_all,
                                                                     const uint32_t* __restrict__ slice_off_all,
                                                                     const uint32_t* __restrict__ full_off_all,
                                                                     const uint32_t* __restrict__ part_list_all,
                                                                     G1RSlot* __restrict__ partial_all, G1RSlot* __restrict__ buckets_all) {
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint32_t* __restrict__ entries = entries_all + (uint64_t)kb * MSM_W * bt.cap_m;
  const uint32_t* __restrict__ offsets = offsets_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t* __restrict__ slice_off = slice_off_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t* __restrict__ full_off = full_off_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t* __restrict__ part_list = part_list_all + (uint64_t)kb * (MSM_NB + 1);
  G1RSlot* __restrict__ partial = partial_all + (uint64_t)kb * bt.cap_slices;
  G1RSlot* __restrict__ buckets = buckets_all + (uint64_t)kb * MSM_NB;
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nfull = full_off[MSM_NB], npart = part_list[MSM_NB];
  // only the lanes of FULL slices search for their bucket; a workgroup that holds none (with 2^19 buckets: almost all of
  // them, a bucket being one partial slice) skips the staging of the search table
#if PLONK_MSM_NB_BITS > 15
  // 2^19 buckets: a bucket is one PARTIAL slice (~24 of 64 entries), full slices only exist for skewed digits — their lanes
  // search full_off in global memory (19 dependent loads, rare) and the workgroup carries no LDS: round 5 removed the 32 KiB
  // staging table every workgroup of this variant reserved for a search almost none of them ran
  if (s >= nfull + npart) return;
  uint32_t b, q, end;
  if (s < nfull) {
    uint32_t lo = 0, hi = MSM_NB / 64 - 1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (full_off[mid * 64] <= s) lo = mid; else hi = mid - 1;
    }
    lo *= 64;
    hi = lo + 63;
#else
  __shared__ uint32_t coarse[MSM_NB / 64];
  if (blockIdx.x * blockDim.x < nfull) {   // uniform per workgroup
    for (uint32_t j = threadIdx.x; j < MSM_NB / 64; j += blockDim.x) coarse[j] = full_off[j * 64];
    __syncthreads();
  }
  if (s >= nfull + npart) return;
  uint32_t b, q, end;
  if (s < nfull) {
    uint32_t lo = 0, hi = MSM_NB / 64 - 1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (coarse[mid] <= s) lo = mid; else hi = mid - 1;
    }
    lo *= 64;
    hi = lo + 63;
#endif
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (full_off[mid] <= s) lo = mid; else hi = mid - 1;
    }
    b = lo;
    q = s - full_off[b];
    end = offsets[b] + (q + 1) * bt.ksl;
  } else {
    b = part_list[s - nfull];
    q = (offsets[b + 1] - offsets[b]) / bt.ksl;
    end = offsets[b + 1];
  }
  const uint32_t beg = offsets[b] + q * bt.ksl;
  G1R acc = G1R::identity();
  uint32_t ent = entries[beg];
  Fp28 x = ld_f28(&table[ent & 0x7fffffffu].x);
  Fp28 y = ld_f28(&table[ent & 0x7fffffffu].y);
  uint32_t k0 = beg;
  ACC_FIRST_PAIR(acc, ent, x, y, k0, end);
  for (uint32_t k = k0; k < end; ++k) {
    const uint32_t ent_c = ent;
    const Fp28 xc = x, yc = y;
    if (k + 1 < end) {
      ent = entries[k + 1];
      x = ld_f28(&table[ent & 0x7fffffffu].x);
      y = ld_f28(&table[ent & 0x7fffffffu].y);
    }
    Fp28 y2;
    const bool neg = ent_c & 0x80000000u;
#pragma unroll
    for (int i = 0; i < Fp28::N; ++i) y2.l[i] = neg ? Fp28::pad<4>(i) - yc.l[i] : yc.l[i];
    acc = acc.add_affine(xc, y2);
  }
  // a bucket that is ONE slice needs no bucket sum: its lane writes the bucket itself (msm_bucket_sum skips it)
  const uint32_t so = slice_off[b];
  if (slice_off[b + 1] - so == 1) st_g1r(buckets + b, acc);
  else st_g1r(partial + so + q, acc);
}

// Occupancy experiment (PLONK_MSM_ACC=lds), kept as the measured answer to "would a third wave per SIMD help?":
// same slices, same additions, THREE waves per SIMD instead of two — and the same time per proof as the kernel above
// (same-box A/B at 2^20: 26.59 / 26.84 ms against 26.84 / 26.93 ms), i.e. msm_accumulate is bound by the number of
// VALU instructions issued, not by latency hiding.  The register-prefetch kernel above needs 216 VGPRs (accumulator 56, current + prefetched table entry 56, P, R, PP,
// PPP, Q 70, a 28-register column accumulator), i.e. two waves per SIMD, where a wave-instruction issues every 4.9
// cycles; at three waves the interval is ~4.6 and at four 4.4 (profiles/r01/valu_issue_rates_gfx950.txt).  Forcing
// 168 registers on that kernel spills 49 of them.  Here (i) the table entry of step k+1 is fetched straight into LDS
// (global_load_lds_dwordx4: no VGPR is held while the gather is in flight; one wave-private 8 KiB slot, chunk j of
// lane L at 1024 j + 16 L, refilled as soon as step k has read it), (ii) the entry words run two steps ahead so that
// the refill never waits for its address, and (iii) the addition is written in the order that keeps the live set
// small (ZZ3 as soon as PP exists, P dead after PPP, X after Q): 167 VGPRs, no scratch, 18 KiB of LDS per workgroup.
typedef __attribute__((address_space(1))) const void* acc_gptr_t;
typedef __attribute__((address_space(3))) void* acc_lptr_t;
__device__ __forceinline__ void acc_prefetch_entry(const G1AffineR* e, uint8_t* wave_slot) {
  const uint8_t* g = reinterpret_cast<const uint8_t*>(e);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    __builtin_amdgcn_global_load_lds((acc_gptr_t)(g + 16 * j), (acc_lptr_t)(wave_slot + 1024 * j), 16, 0, 0);
}
__device__ __forceinline__ Fp28 acc_read_coord(const uint8_t* wave_slot, uint32_t lane, int coord /*0: x, 1: y*/) {
  const uint4* q = reinterpret_cast<const uint4*>(wave_slot + 16 * lane) + 256 * coord;
  const uint4 a = q[0], b = q[64], c = q[128], d = q[192];
  Fp28 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
  r.l[12] = d.x; r.l[13] = d.y;
  return r;
}
__device__ __forceinline__ Fp28 acc_signed_y(const Fp28& y, bool neg) {   // -y as 4p - y, lazy limbs
  Fp28 r;
#pragma unroll
  for (int i = 0; i < Fp28::N; ++i) r.l[i] = neg ? Fp28::pad<4>(i) - y.l[i] : y.l[i];
  return r;
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3)))
msm_accumulate_lds_kernel(const G1AffineR* __restrict__ table, MsmBatch bt, const uint32_t* __restrict__ entries_all,
                          const uint32_t* __restrict__ offsets_all, const uint32_t* __restrict__ slice_off_all,
                          G1RSlot* __restrict__ partial_all) {
  const int kb = (int)blockIdx.y + bt.kb0;
  const uint32_t* __restrict__ entries = entries_all + (uint64_t)kb * MSM_W * bt.cap_m;
  const uint32_t* __restrict__ offsets = offsets_all + (uint64_t)kb * (MSM_NB + 1);
  const uint32_t* __restrict__ slice_off = slice_off_all + (uint64_t)kb * (MSM_NB + 1);
  G1RSlot* __restrict__ partial = partial_all + (uint64_t)kb * bt.cap_slices;
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nslices = slice_off[MSM_NB];
  __shared__ uint32_t coarse[MSM_NB / 64];                          // every 64th slice offset (bucket search, see above)
  __shared__ __attribute__((aligned(16))) uint8_t slot[2][8192];    // one table-entry slot per wave
  for (uint32_t j = threadIdx.x; j < MSM_NB / 64; j += blockDim.x) coarse[j] = slice_off[j * 64];
  __syncthreads();
  if (s >= nslices) return;
  uint8_t* wave_slot = slot[threadIdx.x >> 6];
  const uint32_t lane = threadIdx.x & 63;
  uint32_t lo = 0, hi = MSM_NB / 64 - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (coarse[mid] <= s) lo = mid; else hi = mid - 1;
  }
  lo *= 64;
  hi = lo + 63;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (slice_off[mid] <= s) lo = mid; else hi = mid - 1;
  }
  const uint32_t b = lo;
  const uint32_t beg = offsets[b] + (s - slice_off[b]) * bt.ksl;
  uint32_t end = beg + bt.ksl;
  const uint32_t bend = offsets[b + 1];
  if (end > bend) end = bend;
  G1R acc = G1R::identity();
  uint32_t e_cur = beg < end ? entries[beg] : 0;
  uint32_t e_nxt = beg + 1 < end ? entries[beg + 1] : 0;
  if (beg < end) acc_prefetch_entry(&table[e_cur & 0x7fffffffu], wave_slot);
  for (uint32_t k = beg; k < end; ++k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the slot holds entry k (and e_nxt has arrived)
    const uint32_t ent_c = e_cur;
    const bool neg = ent_c & 0x80000000u;
    if (acc.is_identity()) {   // first entry of the slice (or the step after a cancellation)
      const Fp28 x2 = acc_read_coord(wave_slot, lane, 0);
      const Fp28 y2 = acc_signed_y(acc_read_coord(wave_slot, lane, 1), neg);
      acc = G1R::from_affine(x2, y2.normalized());
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot has been read before it is refilled
      if (k + 1 < end) acc_prefetch_entry(&table[e_nxt & 0x7fffffffu], wave_slot);
      e_cur = e_nxt;
      if (k + 2 < end) e_nxt = entries[k + 2];
      continue;
    }
    // G1R::add_affine (curve28.cuh) with the operands taken from the slot where they are needed; bounds as there
    const Fp28 P_ = Fp28::sub_lazy<32>(Fp28::mul(acc_read_coord(wave_slot, lane, 0), acc.ZZ), acc.X);   // U2 - X  < 34p, lazy
    const Fp28 R_ = Fp28::sub<16>(Fp28::mul(acc_signed_y(acc_read_coord(wave_slot, lane, 1), neg), acc.ZZZ), acc.Y);   // S2 - Y < 18p
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (k + 1 < end) acc_prefetch_entry(&table[e_nxt & 0x7fffffffu], wave_slot);
    e_cur = e_nxt;
    if (k + 2 < end) e_nxt = entries[k + 2];
    if (G1R::maybe_zero(P_) && P_.normalized().is_zero_mod()) {   // same x as the accumulator: double or cancel (rare)
      const G1AffineR* e = &table[ent_c & 0x7fffffffu];             // the slot is being refilled: gather the entry again
      acc = R_.is_zero_mod() ? G1R::dbl_affine(ld_f28(&e->x), acc_signed_y(ld_f28(&e->y), neg).normalized()) : G1R::identity();
      continue;
    }
    const Fp28 PP = P_.sqr();                                 // 34*34 = 1156      -> < 2p
    acc.ZZ = Fp28::mul(acc.ZZ, PP);                           // < 2p
    const Fp28 PPP = Fp28::mul(P_, PP);                       // 34*2              -> < 2p
    const Fp28 Q_ = Fp28::mul(acc.X, PP);                     // 16*2              -> < 2p
    acc.ZZZ = Fp28::mul(acc.ZZZ, PPP);                        // < 2p
    acc.X = Fp28::sub<8>(Fp28::sub_lazy<4>(R_.sqr(), PPP),    // 18*18=324; 2p + 4p
                         Fp28::add_lazy(Q_, Q_));             // - (<4p) + 8p      -> < 14p
    acc.Y = Fp28::mul2(R_, Fp28::sub_lazy<32>(Q_, acc.X),     // 18 * (2+32=34) = 612
                       PPP, Fp28::neg_lazy<16>(acc.Y));       // + 2 * 16 = 644    -> < 2p
  }
  st_g1r(partial + s, acc);
}

// bucket[b] = sum of its slice partials.  Uniform scalars give ~16 slices per bucket: two lanes
// per bucket (strided partial sums + one LDS step) keep the SIMDs busy without idling lanes in a
// deep tree.  Lanes per bucket follow the expected slice count: 1 (sparse), 2 (m ~ 2^20: 16 slices), 4, 8 (m >= 2^22).
// A bucket with more than `heavy_thresh` slices is "heavy" (skewed digits: equal or small scalars — the bits, quads and
// range accumulators of a real witness).  Its lanes here skip it; it is summed in segments of HEAVY_SEG slices (workers
// below), and msm_heavy_bucket adds the segment sums of a bucket of more than one segment: ~14 dependent additions for a
// bucket of any size, and the heavy buckets — which sit next to each other at the small bucket indices — are spread over
// the whole chip.
// With MANY buckets (2^19) the kernel has a third mode: a bucket is one slice whose lane wrote the bucket itself
// (`direct`), and only the listed buckets of 2 .. heavy_thresh slices are visited, a quad each (`multi_list_all`).
static constexpr uint32_t HEAVY_SEG = 128;
// (HeavyItem: plonk_internal.hpp.)  The list of heavy buckets is written by msm_slices_kernel (msm_sort.hip), i.e. BEFORE
// the accumulation, so that the segment sums do not need a kernel of their own: the first `fused` workgroups of
// msm_bucket_sum are segment workers (32 quads: 4 slices per quad + a 5-step tree of quad additions, g1r_add_quad below)
// and run beside the ordinary bucket sums — r03: as separate launches the two heavy kernels cost 0.2 ms per commitment
// group whenever a witness (or the leftover top digit of the bit-position recoding) makes a few buckets heavy.  A bucket
// of one segment goes straight to its bucket slot; longer ones leave segment sums for msm_heavy_bucket.
__device__ __forceinline__ G1R g1r_add_quad(const G1R& a, const G1R& b, uint32_t q);
static constexpr uint32_t HEAVY_FUSED_WGS = MSM_NB_BITS > 15 ? 512 : 128;   // many buckets: the ordinary part of the grid is nearly empty, the skewed low buckets are the kernel
template <int BS_G>   // lanes per bucket, chosen from the expected slices per bucket (msm_batch_device)
__global__ void __launch_bounds__(128) msm_bucket_sum_kernel(MsmBatch bt, const G1RSlot* __restrict__ partial_all,
                                                             const uint32_t* __restrict__ slice_off_all,
                                                             G1RSlot* __restrict__ buckets_all, uint32_t heavy_thresh,
                                                             const uint32_t* __restrict__ nheavy_all, const HeavyItem* __restrict__ heavy_list_all,
                                                             G1RSlot* __restrict__ seg_sum_all, uint64_t seg_cap, uint32_t fused, uint32_t direct,
                                                             const uint32_t* __restrict__ multi_list_all, uint32_t dense_quad) {
  const int kb = (int)blockIdx.y + bt.kb0;
  const G1RSlot* __restrict__ partial = partial_all + (uint64_t)kb * bt.cap_slices;
  const uint32_t* __restrict__ slice_off = slice_off_all + (uint64_t)kb * (MSM_NB + 1);
  G1RSlot* __restrict__ buckets = buckets_all + (uint64_t)kb * MSM_NB;
  __shared__ G1R sh[BS_G > 1 ? 128 : 32];
  if (blockIdx.x < fused) {   // ---- heavy-segment worker
    __shared__ uint32_t found[3];
    const HeavyItem* __restrict__ list = heavy_list_all + (uint64_t)kb * MSM_NB;
    G1RSlot* __restrict__ seg_sum = seg_sum_all + (uint64_t)kb * seg_cap;
    const uint32_t t = threadIdx.x, q = t & 3, L = t >> 2;
    const uint32_t nitems = nheavy_all[2 * kb], nsegs = nheavy_all[2 * kb + 1];
    for (uint32_t sg = blockIdx.x; sg < nsegs; sg += fused) {
      for (uint32_t i = t; i < nitems; i += 128) {   // which item owns segment sg
        const HeavyItem it = list[i];
        if (sg >= it.seg_base && sg < it.seg_base + it.nseg) { found[0] = it.bucket; found[1] = sg - it.seg_base; found[2] = it.nseg; }
      }
      __syncthreads();
      const uint32_t b = found[0], j = found[1], nseg = found[2];
      const uint32_t beg = slice_off[b] + j * HEAVY_SEG, bend = slice_off[b + 1];
      G1R acc = G1R::identity();
      for (uint32_t r = 0; r < HEAVY_SEG / 32; ++r) {
        const uint32_t k = beg + L + 32 * r;
        if (k < bend) acc = g1r_add_quad(acc, ld_g1r(partial + k), q);
      }
      for (uint32_t d = 16; d >= 1; d >>= 1) {
        if (q == 0) sh[L] = acc;
        __syncthreads();
        if (L < d) acc = g1r_add_quad(acc, sh[L + d], q);
        __syncthreads();
      }
      if (t == 0) st_g1r(nseg == 1 ? buckets + b : seg_sum + sg, acc);
      __syncthreads();
    }
    return;
  }
  if (multi_list_all || dense_quad) {
    // many buckets: only the listed buckets of 2 .. heavy_thresh slices (every other one is already written), one lane each,
    // over a SMALL grid with a stride loop — 4096 nearly empty workgroups per commitment cost 0.2 ms of dispatch alone (r03c)
    // A listed bucket has 2 .. heavy_thresh slices, i.e. up to 15 DEPENDENT additions: the kernel lasts as long as its longest
    // chain (0.27 ms with one lane per bucket), so a quad works on each bucket (g1r_add_quad: ~2.8x lower latency).
    // dense_quad (r04, few slices per bucket — small MSMs): the same quad walk over EVERY bucket instead of a list.  With 3-6
    // slices per bucket the one-lane-per-bucket kernel below is 2^15 lanes (half a wave per SIMD) running 3-5 dependent full
    // additions at a lone wave's ~10 cycles per instruction: 200 us for ONE 2^16-term commitment, 400 for a group of four —
    // as long as the accumulation it follows (profiles/r04a).  A quad per bucket is 2^17 lanes and ns - 1 quad additions.
    const uint32_t nmulti = dense_quad ? MSM_NB : nheavy_all[2 * MSM_MAX_BATCH + kb];
    const uint32_t* __restrict__ list = multi_list_all ? multi_list_all + (uint64_t)kb * MSM_NB : nullptr;
    const uint32_t q = threadIdx.x & 3;
    for (uint32_t i = ((blockIdx.x - fused) * blockDim.x + threadIdx.x) >> 2; i < nmulti; i += ((gridDim.x - fused) * blockDim.x) >> 2) {
      const uint32_t mb = dense_quad ? i : list[i];
      const uint32_t beg = slice_off[mb], end = slice_off[mb + 1];
      if (dense_quad) {
        if (end - beg > heavy_thresh || (direct && end - beg == 1)) continue;   // heavy: the segment workers; one slice: written by its lane
        if (end == beg) { if (q == 0) st_g1r(buckets + mb, G1R::identity()); continue; }
      }
      G1R acc = ld_g1r(partial + beg);
      for (uint32_t k = beg + 1; k < end; ++k) acc = g1r_add_quad(acc, ld_g1r(partial + k), q);
      if (q == 0) st_g1r(buckets + mb, acc);
    }
    return;
  }
  const uint32_t t = (blockIdx.x - fused) * blockDim.x + threadIdx.x;
  const uint32_t b = t / BS_G, g = t % BS_G;
  G1R acc = G1R::identity();
  bool heavy = false;
  if (b < MSM_NB) {
    const uint32_t beg = slice_off[b], end = slice_off[b + 1];
    heavy = end - beg > heavy_thresh;   // listed by msm_slices_kernel with the same threshold
    if (direct && end - beg == 1) heavy = true;   // written by its own lane of msm_accumulate_ordered: nothing to do here
    if (!heavy)
      for (uint32_t k = beg + g; k < end; k += BS_G) acc = acc.add(ld_g1r(partial + k));
  }
  for (int d = BS_G / 2; d >= 1; d >>= 1) {
    sh[threadIdx.x] = acc;
    __syncthreads();
    if ((int)g < d) acc = acc.add(sh[threadIdx.x + d]);
    __syncthreads();
  }
  if (g == 0 && b < MSM_NB && !heavy) st_g1r(buckets + b, acc);
}

static constexpr uint32_t HEAVY_WGS = 1024;
__device__ __forceinline__ G1R wg_tree_sum256(G1R acc, G1R* sh) {
  for (int d = 128; d >= 1; d >>= 1) {
    sh[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < d) acc = acc.add(sh[threadIdx.x + d]);
    __syncthreads();
  }
  return acc;
}
// stage 1: workgroup w of a commitment handles segment w, w + HEAVY_WGS, ... of the launch; an item's segments are
// numbered contiguously from its seg_base, so the owner of a segment is found by one pass of the 256 lanes over the
// (short, unordered) item list
__global__ void __launch_bounds__(64) msm_heavy_seg_kernel(MsmBatch bt, const G1RSlot* __restrict__ partial_all,
                                                           const uint32_t* __restrict__ slice_off_all,
                                                           const uint32_t* __restrict__ nheavy_all,
                                                           const HeavyItem* __restrict__ heavy_list_all,
                                                           G1RSlot* __restrict__ seg_sum_all, uint64_t seg_cap) {
  // one WAVE per segment: 4 slices per lane, then a 6-step tree over 14 KiB of LDS — ~11 segments in flight per CU
  // (a 256-lane workgroup with its 57 KiB tree buffer allowed 2, and the segment pass was occupancy-bound)
  const int kb = (int)blockIdx.y + bt.kb0;
  const G1RSlot* __restrict__ partial = partial_all + (uint64_t)kb * bt.cap_slices;
  const uint32_t* __restrict__ slice_off = slice_off_all + (uint64_t)kb * (MSM_NB + 1);
  const HeavyItem* __restrict__ list = heavy_list_all + (uint64_t)kb * MSM_NB;
  G1RSlot* __restrict__ seg_sum = seg_sum_all + (uint64_t)kb * seg_cap;
  __shared__ G1R sh[64];
  __shared__ uint32_t found[2];
  const uint32_t nitems = nheavy_all[2 * kb], nsegs = nheavy_all[2 * kb + 1];
  for (uint32_t sg = blockIdx.x; sg < nsegs; sg += gridDim.x) {
    for (uint32_t i = threadIdx.x; i < nitems; i += 64) {   // which item owns segment sg
      const HeavyItem it = list[i];
      if (sg >= it.seg_base && sg < it.seg_base + it.nseg) { found[0] = it.bucket; found[1] = sg - it.seg_base; }
    }
    __syncthreads();
    const uint32_t b = found[0], j = found[1];
    const uint32_t beg = slice_off[b] + j * HEAVY_SEG, bend = slice_off[b + 1];
    G1R acc = G1R::identity();
    for (uint32_t q = 0; q < HEAVY_SEG / 64; ++q) {
      const uint32_t k = beg + threadIdx.x + 64 * q;
      if (k < bend) acc = acc.add(ld_g1r(partial + k));
    }
    for (int d = 32; d >= 1; d >>= 1) {
      sh[threadIdx.x] = acc;
      __syncthreads();
      if ((int)threadIdx.x < d) acc = acc.add(sh[threadIdx.x + d]);
      __syncthreads();
    }
    if (threadIdx.x == 0) st_g1r(seg_sum + sg, acc);
    __syncthreads();
  }
}
// stage 2: one heavy bucket per workgroup: sum of its segment sums
__global__ void __launch_bounds__(256) msm_heavy_bucket_kernel(const uint32_t* __restrict__ nheavy_all,
                                                               const HeavyItem* __restrict__ heavy_list_all,
                                                               const G1RSlot* __restrict__ seg_sum_all, uint64_t seg_cap,
                                                               G1RSlot* __restrict__ buckets_all, int kb0) {
  const int kb = (int)blockIdx.y + kb0;
  const HeavyItem* __restrict__ list = heavy_list_all + (uint64_t)kb * MSM_NB;
  const G1RSlot* __restrict__ seg_sum = seg_sum_all + (uint64_t)kb * seg_cap;
  G1RSlot* __restrict__ buckets = buckets_all + (uint64_t)kb * MSM_NB;
  __shared__ G1R sh[256];
  const uint32_t nitems = nheavy_all[2 * kb];
  for (uint32_t i = blockIdx.x; i < nitems; i += gridDim.x) {
    const HeavyItem it = list[i];
    G1R acc = G1R::identity();
    for (uint32_t k = threadIdx.x; k < it.nseg; k += 256) acc = acc.add(ld_g1r(seg_sum + it.seg_base + k));
    acc = wg_tree_sum256(acc, sh);
    if (threadIdx.x == 0) st_g1r(buckets + it.bucket, acc);
    __syncthreads();
  }
}

// ---- weighted bucket reduction  W = sum_{b=1..NB} b * B_b  in two shallow kernels -------------
// Write b = 128 h + l with h in [0,256), l in [1,128]  (bucket array index b-1 = 128 h + (l-1)):
//     W = sum_l l * C_l + 128 * sum_h h * R_h ,   C_l = sum_h B_{h,l} ,  R_h = sum_l B_{h,l} .
// msm_rowcol_kernel forms the 256 row sums and 128 column sums (one wave each);
// msm_final_kernel turns both weighted sums into sums of suffix sums (Hillis-Steele scans in LDS),
// doubles the row part 7 times and tree-sums everything: ~24 dependent additions instead of the
// ~70 of a running-sum-per-chunk scheme — these kernels are pure latency (one wave per SIMD).
static constexpr uint32_t RC_ROWS = 256, RC_COLS = 128;
static_assert(MSM_NB_BITS != 15 || RC_ROWS * RC_COLS == MSM_NB, "row/column split must cover the bucket range (2^15 buckets; more buckets fold to this shape)");

// One wave (64 lanes) per sum, four sums per workgroup: a lane adds 2 (rows) or 4 (columns)
// buckets serially, then a 6-step LDS tree — depth 7 / 9 additions and ~1.5 waves per SIMD for a
// group of four commitments, so the whole kernel is one latency round.
static constexpr uint32_t RC_WG_ROWS = RC_ROWS / 4, RC_WG = (RC_ROWS + RC_COLS) / 4;
__global__ void __launch_bounds__(256) msm_rowcol_kernel(const G1RSlot* __restrict__ buckets_all,
                                                         G1RSlot* __restrict__ rc_all) {
  __shared__ G1R sh[256];
  const G1RSlot* __restrict__ buckets = buckets_all + (uint64_t)blockIdx.y * MSM_NB;
  G1RSlot* __restrict__ rc = rc_all + (uint64_t)blockIdx.y * (RC_ROWS + RC_COLS);
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t sum = blockIdx.x * 4 + wave;   // 0..255 rows, 256..383 columns; uniform kind per workgroup
  G1R acc;
  if (blockIdx.x < RC_WG_ROWS) {     // R_h : the 128 buckets of row h
    const G1RSlot* row = buckets + sum * RC_COLS;
    acc = ld_g1r(row + lane).add(ld_g1r(row + lane + 64));
  } else {                           // C_l : the 256 buckets of column l0
    const G1RSlot* col = buckets + (sum - RC_ROWS);
    acc = ld_g1r(col + (uint64_t)lane * RC_COLS);
    for (uint32_t k = 1; k < 4; ++k) acc = acc.add(ld_g1r(col + (uint64_t)(lane + 64 * k) * RC_COLS));
  }
  for (uint32_t d = 32; d >= 1; d >>= 1) {
    sh[t] = acc;
    __syncthreads();
    if (lane < d) acc = acc.add(sh[t + d]);
    __syncthreads();
  }
  if (lane == 0) st_g1r(rc + sum, acc);
}

// one workgroup of 384 lanes per commitment: lanes 0..255 own R_h, lanes 256..383 own C_l
__global__ void __launch_bounds__(384) msm_final_kernel(MsmBatch bt, const G1RSlot* __restrict__ rc_all) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  G1R* sh = reinterpret_cast<G1R*>(smem_raw);          // 384 points
  const G1RSlot* __restrict__ rc = rc_all + (uint64_t)blockIdx.x * (RC_ROWS + RC_COLS);
  G1* __restrict__ out = bt.out[blockIdx.x];
  const uint32_t t = threadIdx.x;
  const bool is_row = t < RC_ROWS;
  const uint32_t seg_end = is_row ? RC_ROWS : (RC_ROWS + RC_COLS);   // suffix scans stay inside the segment
  G1R acc = ld_g1r(rc + t);
  for (uint32_t d = 1; d < RC_ROWS; d <<= 1) {          // inclusive suffix scan: acc_t = sum_{j >= t} x_j
    sh[t] = acc;
    __syncthreads();
    if (t + d < seg_end) acc = acc.add(sh[t + d]);
    __syncthreads();
  }
  // rows: sum_h h R_h = sum_{k=1..255} Suf_k  (drop k = 0), times 128 ; cols: sum_l l C_l = sum_{k=0..127} Suf'_k
  G1R total = G1R::identity();                          // lane 0: Suf_0 = S, the sum of all buckets
  if (is_row) {
    if (t == 0) { total = acc; acc = G1R::identity(); }
    else for (int k = 0; k < 7; ++k) acc = acc.dbl();
  }
  for (uint32_t d = 256; d >= 1; d >>= 1) {             // tree over 384 (< 512) entries
    sh[t] = acc;
    __syncthreads();
    if (t < d && t + d < RC_ROWS + RC_COLS) acc = acc.add(sh[t + d]);
    __syncthreads();
  }
  if (t == 0) {
    if (bt.rows == MSM_ROWS_BITPOS) acc = acc.dbl().add(g1r_neg(total));   // entries weigh 2 b + 1: 2 W - S
    st_g1(out, acc.to_g1());
  }
}

// Alternative tail used by the device prover (prover.hip): instead of finishing W on one
// workgroup (~24 dependent additions at ~18 us each), emit the 16 "bit sums"
//   T_j  = sum of the rows    h whose index has bit j set (j < 8),
//   T'_j = sum of the columns l whose weight has bit j set (j < 7),  and C_128,
// each a plain tree sum of <= 128 points (depth 7, all 16 in parallel), and let the host finish
//   W = sum_j 2^j T'_j + 2^7 C_128 + sum_j 2^(7+j) T_j
// as a 15-term Horner chain: a 64-bit-limb host addition takes ~0.6 us against ~18 us for a
// dependent addition on one GPU lane.  out[k] receives 16 XYZZ points (rows 0..7, cols 8..14, C_128).
__global__ void __launch_bounds__(128) msm_bits_kernel(MsmBatch bt, const G1RSlot* __restrict__ rc_all) {
  __shared__ G1R sh[128];
  const G1RSlot* __restrict__ rc = rc_all + (uint64_t)blockIdx.y * (RC_ROWS + RC_COLS);
  G1* __restrict__ out = bt.out[blockIdx.y] + blockIdx.x;
  const uint32_t u = blockIdx.x, t = threadIdx.x;
  G1R acc = G1R::identity();
  if (u < 8) {                       // 128 of the 256 rows
    const uint32_t h = ((t >> u) << (u + 1)) | (1u << u) | (t & ((1u << u) - 1u));
    acc = ld_g1r(rc + h);
  } else if (u < 15) {               // 64 of the column weights 1..127
    const uint32_t j = u - 8;
    if (t < 64) {
      const uint32_t l = ((t >> j) << (j + 1)) | (1u << j) | (t & ((1u << j) - 1u));
      acc = ld_g1r(rc + RC_ROWS + (l - 1));
    }
  } else if (u == 15) {
    if (t == 0) acc = ld_g1r(rc + RC_ROWS + (RC_COLS - 1));   // weight 128
  } else {                           // u == 16: S = the sum of all 256 rows = of all buckets (bit-position entries: 2 W - S)
    acc = ld_g1r(rc + t).add(ld_g1r(rc + t + 128));
  }
  for (uint32_t d = 64; d >= 1; d >>= 1) {
    sh[t] = acc;
    __syncthreads();
    if (t < d) acc = acc.add(sh[t + d]);
    __syncthreads();
  }
  if (t == 0) st_g1(out, acc.to_g1());
}

// ---- quad-cooperative tail (round 3) -----------------------------------------------------------
// The row/column and bit-sum kernels are chains of DEPENDENT full additions (12 products + 2 squarings,
// ~7.6 k instructions each) on a chip that is mostly idle while they run.  The 14 products of an
// addition form four levels of mutually independent products:
//     U1 U2 S1 S2 | P^2 R^2 ZZ1*ZZ2 ZZZ1*ZZZ2 | P*PP U1*PP ZZ12*PP ZZZ12*P | R*(Q-X3) PPP*(-S1) T*PP
// so the four lanes of a quad hold the SAME two points, each lane computes ONE product per level, and the
// results are exchanged with DPP quad broadcasts (one v_mov_dpp per limb): 4 products + ~0.6 k
// instructions of selects / exchanges instead of 14 products per addition — the latency of an addition drops
// ~2.8x, at 1.6x the issue slots, which these kernels have to spare.  A "logical lane" is a quad; control flow is
// uniform inside a quad by construction (replicated operands), which the DPP reads rely on.
template <int K>
__device__ __forceinline__ Fp28 quad_get(const Fp28& v) {   // lane K of the quad -> all four lanes
  Fp28 r;
#pragma unroll
  for (int i = 0; i < Fp28::N; ++i)
    r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], K * 0x55, 0xf, 0xf, true);
  return r;
}
__device__ __forceinline__ Fp28 quad_sel(uint32_t q, const Fp28& a0, const Fp28& a1, const Fp28& a2, const Fp28& a3) {
  // bit masks, not ?: — the compiler turns a four-way conditional over arrays into a pointer select with the
  // operands parked in scratch memory, which is the last thing a latency-bound chain needs.  Two levels of (m & x) | (~m & y),
  // which is ONE v_bfi_b32 each: 3 instructions per limb (round 6, second session; until then four ANDs and two ORs per
  // limb — 672 of the ~3.3 k instructions of a quad addition were these selects).
  const bool b0 = (q & 1u) != 0, b1 = (q & 2u) != 0;
  Fp28 r;
#pragma unroll
  for (int i = 0; i < Fp28::N; ++i) {
    const uint32_t lo = b0 ? a1.l[i] : a0.l[i], hi = b0 ? a3.l[i] : a2.l[i];   // per-limb selects on two lane masks: v_cndmask_b32
    r.l[i] = b1 ? hi : lo;
  }
  return r;
}
// a + b, both replicated over the quad; q = lane & 3.  Same bounds as G1R::add (Y3 < 4p here: two reductions
// instead of the fused one, inside the Y < 8p contract).
__device__ __forceinline__ G1R g1r_add_quad(const G1R& a, const G1R& b, uint32_t q) {
  if (a.is_identity()) return b;
  if (b.is_identity()) return a;
  const Fp28 m1 = Fp28::mul(quad_sel(q, a.X, b.X, a.Y, b.Y), quad_sel(q, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ));
  const Fp28 U1 = quad_get<0>(m1), U2 = quad_get<1>(m1), S1 = quad_get<2>(m1), S2 = quad_get<3>(m1);   // < 2p
  const Fp28 P_ = Fp28::sub_lazy<4>(U2, U1);                // < 6p, lazy limbs
  const Fp28 R_ = Fp28::sub<4>(S2, S1);                     // < 6p, normalised
  if (G1R::maybe_zero(P_) && P_.normalized().is_zero_mod()) {
    // equal or opposite points (never for independent bucket sums; small or repeated bases reach it): every lane of the
    // quad doubles on its own.  Inlined — an out-of-line call would pin both points in scratch memory on the hot path.
    return R_.is_zero_mod() ? a.dbl() : G1R::identity();
  }
  const Fp28 m2 = Fp28::mul(quad_sel(q, P_, R_, a.ZZ, a.ZZZ), quad_sel(q, P_, R_, b.ZZ, b.ZZZ));   // 6*6, 2*2 -> < 2p
  const Fp28 PP = quad_get<0>(m2), RR = quad_get<1>(m2), ZZ12 = quad_get<2>(m2), ZZZ12 = quad_get<3>(m2);
  const Fp28 m3 = Fp28::mul(quad_sel(q, P_, U1, ZZ12, ZZZ12), quad_sel(q, PP, PP, PP, P_));           // 6*2, 2*2 -> < 2p
  const Fp28 PPP = quad_get<0>(m3), Q_ = quad_get<1>(m3), T_ = quad_get<3>(m3);
  G1R r;
  r.ZZ = quad_get<2>(m3);
  r.X = Fp28::sub<8>(Fp28::sub_lazy<4>(RR, PPP), Fp28::add_lazy(Q_, Q_));                             // < 14p
  const Fp28 m4 = Fp28::mul(quad_sel(q, R_, PPP, T_, T_),                                              // 6*34, 2*4, 2*2 -> < 2p
                            quad_sel(q, Fp28::sub_lazy<32>(Q_, r.X), Fp28::neg_lazy<4>(S1), PP, PP));
  r.Y = Fp28::add(quad_get<0>(m4), quad_get<1>(m4));                                                   // < 4p
  r.ZZZ = quad_get<2>(m4);
  return r;
}

// Row sums R_h (256) and HALF-column sums C'_{half,l} (2 x 128; C_l = C'_{0,l} + C'_{1,l}): 512 sums of 128 buckets each,
// so that every sum has the same depth.  WV waves per sum = 16 WV logical lanes: a logical lane adds 128 / (16 WV)
// buckets serially, then a log2(16 WV)-step LDS tree.  WV is chosen so that one launch fits the chip's wave slots.
static constexpr uint32_t RCQ_SUMS = RC_ROWS + 2 * RC_COLS;
template <int WV>
__global__ void __launch_bounds__(64 * WV) msm_rowcol_quad_kernel(const G1RSlot* __restrict__ buckets_all,
                                                                  G1RSlot* __restrict__ rc_all) {
  constexpr uint32_t LL = 16 * WV;
  __shared__ G1R sh[LL];
  const G1RSlot* __restrict__ buckets = buckets_all + (uint64_t)blockIdx.y * MSM_NB;
  G1RSlot* __restrict__ rc = rc_all + (uint64_t)blockIdx.y * RCQ_SUMS;
  const uint32_t t = threadIdx.x, q = t & 3, L = t >> 2;
  const uint32_t sum = blockIdx.x;
  const G1RSlot* base;
  uint64_t stride;
  if (sum < RC_ROWS) {               // R_h: the 128 buckets of row h
    base = buckets + (uint64_t)sum * RC_COLS;
    stride = 1;
  } else {                           // C'_{half,l}: rows 128 half .. 128 half + 127 of column l
    const uint32_t cidx = sum - RC_ROWS, half = cidx / RC_COLS, l = cidx % RC_COLS;
    base = buckets + (uint64_t)half * (RC_ROWS / 2) * RC_COLS + l;
    stride = RC_COLS;
  }
  G1R acc = ld_g1r(base + (uint64_t)L * stride);
  for (uint32_t k = L + LL; k < 128; k += LL) acc = g1r_add_quad(acc, ld_g1r(base + (uint64_t)k * stride), q);
  for (uint32_t d = LL / 2; d >= 1; d >>= 1) {
    if (q == 0) sh[L] = acc;
    __syncthreads();
    if (L < d) acc = g1r_add_quad(acc, sh[L + d], q);
    __syncthreads();
  }
  if (t == 0) st_g1r(rc + sum, acc);
}

// The 16 bit sums of msm_bits_kernel from the 512 sums above: every bit sum is a plain sum of <= 128 points
// (rows with bit j set; both halves of the 64 columns whose weight has bit j set; the two halves of column 128).
// 64 logical lanes, two points each, a 6-step tree: depth 7.
// rc_stride: sums per commitment in rc; extra = E: the 2^19-bucket variant's 256 "rows" are the folded G_g, its outputs sit E
// slots further (row bits E .. E + 7, columns, C_128, S) and blocks 17 .. 17 + E - 1 add the low row bits from the H_r that
// follow the 2^15 layout in rc (msm_fold_quad_kernel).
__global__ void __launch_bounds__(256) msm_bits_quad_kernel(MsmBatch bt, const G1RSlot* __restrict__ rc_all, uint32_t rc_stride, uint32_t extra) {
  __shared__ G1R sh[64];
  const G1RSlot* __restrict__ rc = rc_all + (uint64_t)blockIdx.y * rc_stride;
  const uint32_t u = blockIdx.x, t = threadIdx.x, q = t & 3, L = t >> 2;
  // S (block 16) is only read for bit-position entries (2 W - S, finish_bit_sums): for window / even-position digits the block
  // — the longest chain of this launch, 256 points against 128 — does not run, and the kernel ends with the bit sums
  if (u == 16 && bt.rows != MSM_ROWS_BITPOS) return;
  G1* __restrict__ out = bt.out[blockIdx.y] + (u < 17 ? u + extra : u - 17);
  G1R acc = G1R::identity();
  for (uint32_t i = L; i < (u == 16 ? 256u : 128u); i += 64) {
    G1R p = G1R::identity();
    if (u >= 17) {                     // low row bit j = u - 17: the parts P[sg][r] (16 per r, msm_fold_quad_kernel) of the H_r whose index r has bit j set
      const uint32_t j = u - 17, hb = extra - 1u;                       // 16 * 2^(E-1) points, at most 128
      if (i < (16u << hb)) {
        const uint32_t sg = i >> hb, rr = i & ((1u << hb) - 1u);
        const uint32_t r = ((rr >> j) << (j + 1)) | (1u << j) | (rr & ((1u << j) - 1u));
        p = ld_g1r(rc + RCQ_SUMS + (sg << extra) + r);
      }
    } else if (u == 16) {              // S = the sum of all 256 rows = of all buckets (bit-position entries: 2 W - S)
      p = ld_g1r(rc + i);
    } else if (u < 8) {                // 128 of the 256 rows
      const uint32_t h = ((i >> u) << (u + 1)) | (1u << u) | (i & ((1u << u) - 1u));
      p = ld_g1r(rc + h);
    } else if (u < 15) {               // 64 of the column weights 1..127, both halves
      const uint32_t j = u - 8, half = i >> 6, tt = i & 63;
      const uint32_t l = ((tt >> j) << (j + 1)) | (1u << j) | (tt & ((1u << j) - 1u));
      p = ld_g1r(rc + RC_ROWS + half * RC_COLS + (l - 1));
    } else if (u == 15 && i < 2) {     // weight 128
      p = ld_g1r(rc + RC_ROWS + i * RC_COLS + (RC_COLS - 1));
    }
    acc = g1r_add_quad(acc, p, q);
  }
  for (uint32_t d = 32; d >= 1; d >>= 1) {
    if (q == 0) sh[L] = acc;
    __syncthreads();
    if (L < d) acc = g1r_add_quad(acc, sh[L + d], q);
    __syncthreads();
  }
  if (t == 0) st_g1(out, acc.to_g1());
}

// The segment sums of one heavy bucket per workgroup, with quad additions (the segments themselves are summed by the fused
// workers of msm_bucket_sum); the list of heavy buckets is short and this is a chain of dependent additions like the two above.
__global__ void __launch_bounds__(256) msm_heavy_bucket_quad_kernel(const uint32_t* __restrict__ nheavy_all,
                                                                    const HeavyItem* __restrict__ heavy_list_all,
                                                                    const G1RSlot* __restrict__ seg_sum_all, uint64_t seg_cap,
                                                                    G1RSlot* __restrict__ buckets_all, uint32_t fused, int kb0) {
  const int kb = (int)blockIdx.y + kb0;
  const HeavyItem* __restrict__ list = heavy_list_all + (uint64_t)kb * MSM_NB;
  const G1RSlot* __restrict__ seg_sum = seg_sum_all + (uint64_t)kb * seg_cap;
  G1RSlot* __restrict__ buckets = buckets_all + (uint64_t)kb * MSM_NB;
  __shared__ G1R sh[64];
  const uint32_t t = threadIdx.x, q = t & 3, L = t >> 2;
  const uint32_t nitems = nheavy_all[2 * kb];
  for (uint32_t i = blockIdx.x; i < nitems; i += gridDim.x) {
    const HeavyItem it = list[i];
    if (it.nseg == 1 && fused) continue;   // the fused segment worker of msm_bucket_sum wrote the bucket itself (uniform per workgroup)
    G1R acc = G1R::identity();
    for (uint32_t k = L; k < it.nseg; k += 64) acc = g1r_add_quad(acc, ld_g1r(seg_sum + it.seg_base + k), q);
    for (uint32_t d = 32; d >= 1; d >>= 1) {
      if (q == 0) sh[L] = acc;
      __syncthreads();
      if (L < d) acc = g1r_add_quad(acc, sh[L + d], q);
      __syncthreads();
    }
    if (t == 0) st_g1r(buckets + it.bucket, acc);
    __syncthreads();
  }
}

}  // namespace PLONK_MSM_NS

#if PLONK_MSM_NB_BITS == 15   // ---- shared ----
__global__ void xyzz_to_affine97_kernel(const G1* __restrict__ in, uint8_t* __restrict__ out97) {
  if (threadIdx.x != 0) return;
  G1Affine a;
  const bool finite = ld_g1(in).to_affine(&a);
  uint32_t* o = reinterpret_cast<uint32_t*>(out97);
#pragma unroll
  for (int k = 0; k < 12; ++k) { o[k] = a.x.l[k]; o[12 + k] = a.y.l[k]; }
  out97[96] = finite ? 0 : 1;
}

__global__ void msm_identity_kernel(G1* out) {
  if (threadIdx.x == 0) st_g1(out, G1::identity());
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Rows of the tables of an n-point key.  A row per bit position (256 rows, 32 KiB per point) saves 8 % of the additions of
// every MSM over the key (25 % with the 2^19-bucket layout it also enables); a row per SECOND bit position (128 rows, round
// 4: for_each_digit_even, 12.8 instead of 16 additions per scalar) is half the memory.  Which one a key gets follows the
// context's TABLE BUDGET (plonk_gpu_config.table_budget_bytes, default 80 % of the device's total memory) and what the
// context already holds — not the memory that happens to be free (until round 4: "half of what hipMemGetInfo reports free
// right now", so a neighbour process or a second context silently changed the layout, the additions per scalar and the
// run time):
//   * the COMMIT key is loaded before the prover's buffers and the Lagrange-basis key exist and leaves room for them
//     (ADVICE r4): bit-position rows when they take at most 60 % of the budget, every second position at most 30 %;
//   * a prover's LAGRANGE-basis key is built last, nothing grows after it: the densest layout that fits in what is left.
// 2^20 points: 32 GiB per key, twice per prover context, of 288 GB; 2^22: 137 GB for the commit key + 69 GB of half-density
// rows for the Lagrange-basis key.  plonk_gpu_config.table_mode (PLONK_MSM_TABLE=window|halfpos|bitpos) forces a layout.
uint64_t msm_table_bytes(uint32_t rows, uint64_t n) {
  const uint64_t scratch = sizeof(Fp28Slot) * 3 * (uint64_t)(rows - 1) * (1ull << 16);   // srs_table_kernel's per-chunk scratch
  return sizeof(G1AffineR) * (uint64_t)rows * n + (rows > MSM_ROWS_WINDOW ? scratch : 0);
}
uint32_t msm_table_rows(const Ctx* c, uint64_t n, bool last_key) {
  auto index_ok = [n](uint32_t rows) { return (uint64_t)rows * n <= (1ull << 31); };   // 31-bit table index of an entry
  const int mode = c->cfg.table_mode;
  if (mode == (int)MSM_ROWS_WINDOW) return MSM_ROWS_WINDOW;
  if (mode == (int)MSM_ROWS_HALFPOS) return MSM_ROWS_HALFPOS;                          // (table_index_check refuses an oversized key)
  if (mode == (int)MSM_ROWS_BITPOS) return index_ok(MSM_ROWS_BITPOS) ? MSM_ROWS_BITPOS : MSM_ROWS_WINDOW;
  // small keys: the saved additions do not pay for the longer recoding, the skewed top digit and — with the 2^19-bucket
  // layout — a reduction over 16x the buckets (same-box pairs, window rows vs bit-position rows: 2^16 gates 5.33 / 5.59 ms,
  // 2^18 11.16 / 11.37, 2^19 20.23 / 20.43; 2^20 37.8 / 34.9, 2^21 71.9* / 64.4, 2^22 147.0 / 131.8; * = 2^15 buckets).
  // profiles/r03c/sizes_2p18_to_2p22.txt
  // round 4: keys of 2^18 .. 2^19 points take bit-position rows too (2^19 buckets win from 2^19 terms on, see msm_batch_device)
  if (n <= (1ull << 18) + 64) return MSM_ROWS_WINDOW;
  const uint64_t budget = c->cfg.table_budget;
  const uint64_t left = budget > c->table_bytes ? budget - c->table_bytes : 0;
  // the 31-bit index limit is checked per layout (keys of 2^23 .. 2^24 points can still take half-density rows)
  const uint32_t layouts[2] = {MSM_ROWS_BITPOS, MSM_ROWS_HALFPOS};
  const uint64_t share[2] = {budget / 10 * 6, budget / 10 * 3};
  for (int k = 0; k < 2; ++k) {
    if (!index_ok(layouts[k])) continue;
    const uint64_t need = msm_table_bytes(layouts[k], n);
    if (need <= left && (last_key || need <= share[k])) return layouts[k];
  }
  return MSM_ROWS_WINDOW;
}
static int table_index_check(uint32_t rows, uint64_t n) {
  if ((uint64_t)rows * n > (1ull << 31)) return (plonk::set_last_error("invalid argument", "commit key: table rows * points must be <= 2^31 (31-bit table index of an entry)", __FILE__, __LINE__), PLONK_ERR_ARG);
  return PLONK_OK;
}
// allocate the tables of an n-point key (replacing the old key); the rows are filled by srs_table_chunk
int srs_table_begin(Ctx* c, uint64_t n) {
  ++c->srs_gen;   // provers built on the previous key refuse to prove (prover.hip)
  if (c->srs_table) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(c->srs_table));
    c->table_bytes -= c->srs_table_alloc < c->table_bytes ? c->srs_table_alloc : c->table_bytes;
    c->srs_table_alloc = 0;
    c->srs_table = nullptr; c->srs_n = 0; c->srs_rows = 0;
  }
  if (n == 0) return PLONK_OK;
  const uint32_t rows = msm_table_rows(c, n, false);
  { const int rc = table_index_check(rows, n); if (rc) return rc; }
  HIP_TRY(hipMalloc((void**)&c->srs_table, sizeof(G1AffineR) * (size_t)rows * n));
  c->srs_rows = rows;
  c->srs_table_alloc = sizeof(G1AffineR) * (uint64_t)rows * n;   // (callers set srs_n once the rows are filled; the bytes are held from here on)
  c->table_bytes += c->srs_table_alloc;
  return PLONK_OK;
}
// scratch of srs_table_kernel, kept on the context between the chunks of one load
static int srs_table_scratch(Ctx* c, uint64_t slots) {
  if (c->table_scratch_pts >= slots) return PLONK_OK;
  if (c->table_scratch) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->table_scratch)); c->table_scratch = nullptr; c->table_scratch_pts = 0; }
  HIP_TRY(hipMalloc(&c->table_scratch, sizeof(Fp28Slot) * slots));
  c->table_scratch_pts = slots;   // capacity in 64-byte slots
  return PLONK_OK;
}
void srs_table_scratch_free(Ctx* c) {   // after the stream that ran the table kernels was synchronised
  if (c->table_scratch) (void)hipFree(c->table_scratch);
  c->table_scratch = nullptr;
  c->table_scratch_pts = 0;
}
// table entries of points [first, first + count), read from pts_dev[0 .. count), on `st` (always the context's main
// stream: the launches share one scratch array and rely on stream order)
static int srs_table_launch(Ctx* c, const G1Affine* pts_dev, G1AffineR* table, uint32_t rows, uint64_t n, uint64_t first, uint64_t count, hipStream_t st) {
  const uint64_t chunk = srs_table_chunk_points(rows);
  const uint32_t step = rows == MSM_ROWS_BITPOS ? 1u : (rows == MSM_ROWS_HALFPOS ? 2u : (uint32_t)MSM_C);
  for (uint64_t off = 0; off < count; off += chunk) {
    const uint64_t cnt = count - off < chunk ? count - off : chunk;
    const int rc = srs_table_scratch(c, 3ull * (rows - 1) * cnt);
    if (rc) return rc;
    hipLaunchKernelGGL(srs_table_kernel, dim3((uint32_t)((cnt + 63) / 64)), dim3(64), 0, st, pts_dev + off, table, n, first + off, cnt,
                       (Fp28Slot*)c->table_scratch, rows, step);
    HIP_TRY(hipGetLastError());
  }
  return PLONK_OK;
}
int srs_table_chunk(Ctx* c, const G1Affine* pts_dev, uint64_t n, uint64_t first, uint64_t count, hipStream_t st) {
  if (!count) return PLONK_OK;
  return srs_table_launch(c, pts_dev, (G1AffineR*)c->srs_table, c->srs_rows, n, first, count, st);
}
int srs_table_build(Ctx* c, const G1Affine* pts_dev, uint64_t n, void** table_out, uint32_t* rows_out) {
  *table_out = nullptr;
  *rows_out = 0;
  if (n == 0) return PLONK_OK;
  const uint32_t rows = msm_table_rows(c, n, true);
  { const int rc = table_index_check(rows, n); if (rc) return rc; }
  G1AffineR* t = nullptr;
  HIP_TRY(hipMalloc((void**)&t, sizeof(G1AffineR) * (size_t)rows * n));
  int rc = srs_table_launch(c, pts_dev, t, rows, n, 0, n, c->stream);
  if (rc == PLONK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = PLONK_ERR_HIP;
  srs_table_scratch_free(c);
  if (rc) { (void)hipFree(t); return rc; }
  *table_out = t;
  *rows_out = rows;
  c->table_bytes += sizeof(G1AffineR) * (uint64_t)rows * n;   // released by srs_table_release (prover_free)
  return PLONK_OK;
}
void srs_table_release(Ctx* c, void* table, uint32_t rows, uint64_t n) {
  if (!table) return;
  (void)hipFree(table);
  const uint64_t held = sizeof(G1AffineR) * (uint64_t)rows * n;
  c->table_bytes -= held < c->table_bytes ? held : c->table_bytes;
}

// ---------------------------------------------------------------------------
// Lagrange-basis commit key: [L_i(tau)] G = 1/n * sum_j w^(-ij) [tau^j] G — an inverse FFT over the group.
// In-place radix-2 decimation-in-frequency stages on XYZZ points (natural order in, bit-reversed out): the
// butterfly (a, b) -> (a + b, (a - b) * w^-e) costs one 255-bit double-and-add; n/2 * log2 n of them, once per
// prover.  The last kernel undoes the bit reversal, multiplies by 1/n, normalises to affine and appends the two
// points the blinding terms of a wire polynomial need: [tau^n] G - G and [tau^(n+1)] G - [tau] G.
// ---------------------------------------------------------------------------
// [k] p for a full-width scalar: 128 interleaved double-and-add steps over {p, phi(p), p + phi(p)} (curve28.cuh; until the
// second session of round 6 a 255-step double-and-add: ecfft_stage_kernel 41.5 ms per stage at 2^20 points)
__device__ __forceinline__ G1R g1r_mul_fr(const G1R& p, const Fr& k_canonical) { return g1r_mul_glv(p, k_canonical.l); }
__global__ void ecfft_load_kernel(const G1AffineR* __restrict__ row0, G1RSlot* __restrict__ v, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st_g1r(v + i, G1R::from_affine(ld_f28(&row0[i].x), ld_f28(&row0[i].y)));
}
// stage with butterfly span `half`: pairs (blk * 2 half + j, + half), twiddle w_inv^(j * n / (2 half))
__global__ void __launch_bounds__(64) ecfft_stage_kernel(G1RSlot* __restrict__ v, uint64_t n, uint64_t half, Fr w_inv) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / 2) return;
  const uint64_t j = t % half, lo = (t / half) * 2 * half + j, hi = lo + half;
  const G1R a = ld_g1r(v + lo), b = ld_g1r(v + hi);
  st_g1r(v + lo, a.add(b));
  G1R d = a.add(g1r_neg(b));
  if (j) d = g1r_mul_fr(d, w_inv.pow_u64(j * (n / (2 * half))).from_mont());
  st_g1r(v + hi, d);
}
__global__ void __launch_bounds__(64) ecfft_finish_kernel(const G1RSlot* __restrict__ v, const G1AffineR* __restrict__ row0, uint64_t n,
                                                          uint32_t L, Fr n_inv_canonical, G1Affine* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + 2) return;
  G1R p;
  if (i < n) {
    uint64_t r = 0;
    for (uint32_t b = 0; b < L; ++b) r |= ((i >> b) & 1) << (L - 1 - b);
    p = g1r_mul_fr(ld_g1r(v + r), n_inv_canonical);
  } else {   // [tau^(n + k)] G - [tau^k] G, k = i - n
    const uint64_t k = i - n;
    p = G1R::from_affine(ld_f28(&row0[n + k].x), ld_f28(&row0[n + k].y))
            .add(g1r_neg(G1R::from_affine(ld_f28(&row0[k].x), ld_f28(&row0[k].y))));
  }
  G1Affine a;
  if (p.is_identity()) {   // cannot happen for a key from a real setup (tau^n != 1); keep the slot defined
    for (int k = 0; k < 12; ++k) { a.x.l[k] = 0; a.y.l[k] = 0; }
  } else {
    Fp28 x, y;
    g1r_to_affine(p, &x, &y);
    a.x = x.to_fp();
    a.y = y.to_fp();
  }
  st_aff(out + i, a);
}

// out[j] = [tau^(n + k0 + j)] G - [tau^(k0 + j)] G for j < cnt: the points the blinding terms b_k X^k (X^n - 1) of a blinded
// polynomial commit over (k = 0, 1 for the wires — part of the Lagrange-basis key — and k = 2 for z, prover.hip)
__global__ void lagrange_blind_points_kernel(const G1AffineR* __restrict__ row0, uint64_t n, uint32_t k0, uint32_t cnt, G1Affine* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  const uint64_t k = k0 + j;
  const G1R p = G1R::from_affine(ld_f28(&row0[n + k].x), ld_f28(&row0[n + k].y))
                    .add(g1r_neg(G1R::from_affine(ld_f28(&row0[k].x), ld_f28(&row0[k].y))));
  G1Affine a;
  if (p.is_identity()) {
    for (int i = 0; i < 12; ++i) { a.x.l[i] = 0; a.y.l[i] = 0; }
  } else {
    Fp28 x, y;
    g1r_to_affine(p, &x, &y);
    a.x = x.to_fp();
    a.y = y.to_fp();
  }
  st_aff(out + j, a);
}
int lagrange_blind_points_device(Ctx* c, uint64_t n, uint32_t k0, uint32_t cnt, G1Affine* out_dev) {
  if (!c->srs_table || c->srs_n < n + k0 + cnt) return (plonk::set_last_error("invalid argument", "blinding points need size + k commit-key points", __FILE__, __LINE__), PLONK_ERR_DEGREE);
  hipLaunchKernelGGL(lagrange_blind_points_kernel, dim3(1), dim3(64), 0, c->stream, (const G1AffineR*)c->srs_table, n, k0, cnt, out_dev);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

int lagrange_points_device(Ctx* c, uint32_t L, G1Affine* out_dev) {
  const uint64_t n = 1ull << L;
  if (!c->srs_table || c->srs_n < n + 2) return (plonk::set_last_error("invalid argument", "Lagrange key needs size + 2 commit-key points", __FILE__, __LINE__), PLONK_ERR_DEGREE);
  hipStream_t st = c->stream;
  const G1AffineR* row0 = (const G1AffineR*)c->srs_table;   // window 0 = the key points themselves
  G1RSlot* v = nullptr;
  HIP_TRY(hipMalloc((void**)&v, sizeof(G1RSlot) * n));
  hipLaunchKernelGGL(ecfft_load_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, row0, v, n);
  Fr w = fr_root_of_unity();
  for (uint32_t i = L; i < 32; ++i) w = w.sqr();
  const Fr w_inv = w.inv();
  for (uint64_t half = n / 2; half >= 1; half >>= 1)
    hipLaunchKernelGGL(ecfft_stage_kernel, dim3((uint32_t)((n / 2 + 63) / 64)), dim3(64), 0, st, v, n, half, w_inv);
  const Fr n_inv = Fr::from_u64(n).inv().from_mont();
  hipLaunchKernelGGL(ecfft_finish_kernel, dim3((uint32_t)((n + 2 + 63) / 64)), dim3(64), 0, st, (const G1RSlot*)v, row0, n, L, n_inv, out_dev);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(v);
  if (e != hipSuccess) { set_last_error("lagrange_points_device", hipGetErrorString(e), __FILE__, __LINE__); return PLONK_ERR_HIP; }
  return PLONK_OK;
}

int srs_load_device(Ctx* c, const G1Affine* pts_dev, uint64_t n) {
  int rc = srs_table_begin(c, n);
  if (rc || n == 0) return rc;
  rc = srs_table_chunk(c, pts_dev, n, 0, n, c->stream);
  if (rc == PLONK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = PLONK_ERR_HIP;
  srs_table_scratch_free(c);
  if (rc) return rc;
  c->srs_n = n;
  return PLONK_OK;
}

// the commit key back in the ABI's form (x || y, 12 x 32-bit Montgomery limbs each, canonical) from row 0 of the
// window tables: what CommitKey::to_raw_var_bytes serialises (key.rs:215-229) — plonk_prover_to_bytes
__global__ void srs_export_kernel(const G1AffineR* __restrict__ row0, uint64_t n, G1Affine* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine a;
  a.x = ld_f28(&row0[i].x).to_fp();
  a.y = ld_f28(&row0[i].y).to_fp();
  st_aff(out + i, a);
}
int srs_export_device(Ctx* c, G1Affine* out_dev) {
  if (!c->srs_n) return PLONK_OK;
  hipLaunchKernelGGL(srs_export_kernel, dim3((uint32_t)((c->srs_n + 63) / 64)), dim3(64), 0, c->stream,
                     (const G1AffineR*)c->srs_table, c->srs_n, out_dev);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

int srs_validate_device(Ctx* c, const G1Affine* pts_dev, uint64_t n, int* flag_dev) {
  HIP_TRY(hipMemsetAsync(flag_dev, 0, sizeof(int), c->stream));
  if (n) hipLaunchKernelGGL(srs_validate_kernel, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, c->stream, pts_dev, n, flag_dev);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}

int srs_generate_device(Ctx* c, const Fr& tau, const Fr& g_scalar, uint64_t n, G1Affine* out_dev) {
  hipLaunchKernelGGL(srs_generate_kernel, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, c->stream, tau, g_scalar, n, out_dev);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  return PLONK_OK;
}

#endif   // shared

// ---------------------------------------------------------------------------
// host side of the MSM
// ---------------------------------------------------------------------------
void prof_begin(Ctx* c, int slot);
void prof_end(Ctx* c, int slot);

namespace PLONK_MSM_NS {

// Slice length.  2^15 buckets: 32 entries from m = 2^20 up; halved with m below that (down to 4) so that a smaller MSM
// still spreads over ~2^19 lanes instead of leaving most SIMDs idle behind 32 serial additions.  2^19 buckets: always 32 —
// a bucket holds ~24 entries at m = 2^20, i.e. one lane per bucket (in order of length) and almost no second slices.
uint32_t msm_ksl(const Ctx* c, uint64_t m) {
  const int forced = c->cfg.ksl;   // PLONK_MSM_KSL: tuning experiments only
  if (forced == 4 || forced == 8 || forced == 16 || forced == 32 || forced == 64 || forced == 128) return (uint32_t)forced;
  // 2^17 buckets (r04): 32-entry slices although a bucket holds ~53 entries at 2^19 terms.  One lane per bucket (128-entry
  // slices) is a single round of 2^17 lanes per commitment whose longest buckets set the kernel's time: accumulate ran at
  // 83 % of its rate (13.4 ms per 2^19-gate proof against 11.2 with 32-entry slices, profiles/r04f); the second slice of
  // every bucket costs 0.9 ms of bucket sums.
  if (MSM_NB_BITS == 17) return 32u;
  if (MSM_NB_BITS > 15) {   // one slice per bucket: the smallest of 32 / 64 / 128 that holds ~1.3x the expected entries of a bucket
    const uint64_t expect = 13 * m / MSM_NB;
    return expect <= 24 ? 32u : (expect <= 49 ? 64u : 128u);
  }
  // smallest power of two >= 3 m / 2^15, clamped to [4, 32]: 3-6 slices per bucket.  (Until r03e the rule was m / 2^15,
  // 4-8 slices: same-box A/B 2^16 5.13 / 5.11 -> 4.97 / 4.94 ms, 2^19 20.36 / 19.72 -> 20.24 / 19.50, others unchanged.)
  uint32_t r = 4;
  while (r < MSM_KSL && (uint64_t)r * MSM_NB < 3 * m) r *= 2;
  return r;
}

// lanes in order of slice length (msm_slices_kernel<true> + msm_accumulate_ordered_kernel): the default from 16-entry slices on.
// Until round 6 from 32-entry slices on (m > 2^17.4).  Same-box A/B at 16-entry slices, three repetitions
// (profiles/r06b/order17.jsonl): a 2^17-gate proof 6.69 / 6.69 / 6.70 -> 6.57 / 6.52 / 6.50 ms (accumulate 3.57 -> 3.39), a
// rank of 8 alone at 2^20 gates (2^17-point slices) 6.66 / 6.63 / 6.59 -> 6.60 / 6.53 / 6.59; at 8-entry slices (2^16
// gates) no difference (4.62-4.68 either way), so shorter slices keep the bucket order and save the ordering launch.
bool msm_acc_ordered(const Ctx* c, uint32_t ksl) { return c->cfg.order >= 0 ? c->cfg.order == 1 : ksl >= 16; }

#if PLONK_MSM_NB_BITS > 15
// ---- reduction tail for MANY buckets: throughput first ---------------------------------------------------------------
// W = sum_b (b + 1) B_b over 2^19 buckets = 4096 rows x 128 columns.  With 16x the buckets of the 2^15 layout the row /
// column sums are no longer a latency problem but 2 x 2^19 full additions per commitment: stage 1 runs them at full lane
// efficiency (a lane adds 16 buckets serially, 8 lanes finish a sum of 128 with a 3-step tree):
//     R_h          = sum of row h                         (ROWS sums)
//     CP[p][l]     = sum over the 128 rows of part p of column l   (PARTS x 128 sums)
// Stage 1.5 folds them to the 2^15 layout's shapes — row index h = 2^E g + r:  G_g = sum_r R_h (256 sums), H_r = sum_g R_h
// (2^E sums), C_l = sum_p CP[p][l] (128 sums) — and the bit sums follow: row bits >= E from the G_g exactly as the 2^15
// variant takes them from its 256 rows (msm_bits_quad_kernel, outputs shifted by E), row bits < E from the H_r.
static constexpr uint32_t TP_E = MSM_NB_BITS - 15;
static constexpr uint32_t TP_ROWS = MSM_NB / 128, TP_PARTS = TP_ROWS / 128;
static constexpr uint32_t TP_RC1 = 2 * TP_ROWS;                  // stage-1 sums per commitment
static constexpr uint32_t TP_NP = 16u << TP_E;                  // partial low-row sums P[sg][r] = sum of R_h over the 16 groups g of super-group sg, h = 2^E g + r
static constexpr uint32_t TP_RC2 = RCQ_SUMS + TP_NP;            // stage-1.5 sums per commitment: the 2^15 layout + the P[sg][r]
// LPS lanes per sum of 128 buckets: a lane adds 128 / LPS buckets serially, then a log2(LPS)-step tree.  8 for groups of 2-4
// commitments (15 + 3 additions per lane, 88 % of the lane-steps useful), 16 for a single commitment (7 + 4, 72 %): the rule
// and its measurements are at the launch (msm_batch_device_v).
// The LPS partial sums of a sum are combined by QUADS (QT, round 4): the tree of the lane version runs log2(LPS) full
// additions on every wave with half, a quarter, ... of its lanes active; here the 32 quads of the workgroup share the
// PER * (LPS - 1) pair additions level by level (g1r_add_quad: ~2.7 k instructions against ~7.6 k), 4 quad additions in
// sequence for LPS = 8 instead of 3 full ones.  slot(sum, k): where partial k of a sum sits in sh (rows: the lanes of a sum
// are adjacent; columns: PER apart).
template <int LPS, bool QT>
__global__ void __launch_bounds__(128) msm_rowcol_tp_kernel(const G1RSlot* __restrict__ buckets_all, G1RSlot* __restrict__ rc1_all) {
  constexpr uint32_t PER = 128 / LPS;          // buckets per lane = sums per workgroup
  __shared__ G1R sh[128];
  const G1RSlot* __restrict__ buckets = buckets_all + (uint64_t)blockIdx.y * MSM_NB;
  G1RSlot* __restrict__ rc1 = rc1_all + (uint64_t)blockIdx.y * TP_RC1;
  const uint32_t t = threadIdx.x;
  const bool rows = blockIdx.x < TP_ROWS / PER;   // uniform per workgroup
  uint32_t s, out;
  G1R acc;
  if (rows) {                                  // PER rows per workgroup, LPS lanes per row
    const uint32_t row = PER * blockIdx.x + t / LPS;
    s = t % LPS;
    out = row;
    const G1RSlot* base = buckets + (uint64_t)row * 128 + PER * s;
    acc = ld_g1r(base);
    for (uint32_t k = 1; k < PER; ++k) acc = acc.add(ld_g1r(base + k));
  } else {                                     // (part, PER columns) per workgroup, LPS lanes (PER rows each) per column
    const uint32_t w = blockIdx.x - TP_ROWS / PER, p = w / LPS, l0 = PER * (w % LPS) + (t % PER);
    s = t / PER;
    out = TP_ROWS + p * 128 + l0;
    const G1RSlot* base = buckets + ((uint64_t)128 * p + PER * s) * 128 + l0;
    acc = ld_g1r(base);
    for (uint32_t k = 1; k < PER; ++k) acc = acc.add(ld_g1r(base + (uint64_t)k * 128));
  }
  const uint32_t stride = rows ? 1u : PER;     // distance in sh between partial k and k + 1 of the same sum
  if (!QT) {
    for (uint32_t d = LPS / 2; d >= 1; d >>= 1) {
      sh[t] = acc;
      __syncthreads();
      if (s < d) acc = acc.add(sh[t + stride * d]);
      __syncthreads();
    }
    if (s == 0) st_g1r(rc1 + out, acc);
    return;
  }
  sh[t] = acc;
  __syncthreads();
  const uint32_t q = t & 3, quad = t >> 2;
  for (uint32_t m = LPS; m > 1; m >>= 1) {     // m partials per sum -> m / 2
    const uint32_t half = m / 2, adds = PER * half;
    for (uint32_t a = quad; a < adds; a += 32) {
      const uint32_t sum = a / half, k = a % half;
      const uint32_t i0 = rows ? sum * LPS + k : k * PER + sum;
      const G1R r = g1r_add_quad(sh[i0], sh[i0 + stride * half], q);
      if (q == 0) sh[i0] = r;                  // (distinct additions of a level touch distinct slots)
    }
    __syncthreads();
  }
  if (s == 0) st_g1r(rc1 + out, sh[t]);        // partial 0 of the thread's own sum: slot t
}
// stage 1.5 (quad additions): rc2 = [G_g (256) | C_l (128) | identity (128) | P[sg][r] (16 x 2^E)]
// One WAVE per sum (16 logical lanes; round 6, second session).  A sum has 16 points (G_g, P[sg][r]) or 32 (C_l); until then
// a sum was a 256-thread workgroup of 64 logical lanes of which 16 or 32 held a point, running a 6-step tree with two
// workgroup barriers per step: 2560 four-wave workgroups per group of four commitments at two per CU — five rounds of the
// chip, 150-250 us per launch at 2^20 gates for ~10 k quad additions.  A wave per sum is a quarter of the waves, a 4-step
// tree and no workgroup barriers to wait at.
static constexpr uint32_t FOLD_LL = 16;   // logical lanes (quads) per sum
__global__ void __launch_bounds__(64) msm_fold_quad_kernel(const G1RSlot* __restrict__ rc1_all, G1RSlot* __restrict__ rc2_all) {
  __shared__ G1R sh[FOLD_LL];
  const G1RSlot* __restrict__ rc1 = rc1_all + (uint64_t)blockIdx.y * TP_RC1;
  G1RSlot* __restrict__ rc2 = rc2_all + (uint64_t)blockIdx.y * TP_RC2;
  const uint32_t u = blockIdx.x, t = threadIdx.x, q = t & 3, L = t >> 2;
  uint32_t npts, first, stride, dst;
  if (u < 256) { npts = 1u << TP_E; first = u << TP_E; stride = 1; dst = u; }                                   // G_g
  else if (u < 256 + TP_NP) {   // P[sg][r]: 16 of the 256 terms of H_r = sum_g R_(2^E g + r) — a whole H_r was 4 serial + 6 tree quad additions,
    const uint32_t v = u - 256, sg = v >> TP_E, r = v & ((1u << TP_E) - 1u);   // the longest chain of this kernel; the bit sums add the 16 parts (r04)
    npts = 16; first = ((16u * sg) << TP_E) + r; stride = 1u << TP_E; dst = RCQ_SUMS + v;
  }
  else { const uint32_t l0 = u - 256 - TP_NP; npts = TP_PARTS; first = TP_ROWS + l0; stride = 128; dst = RC_ROWS + l0; }   // C_l
  G1R acc = G1R::identity();
  for (uint32_t i = L; i < npts; i += FOLD_LL) acc = g1r_add_quad(acc, ld_g1r(rc1 + first + (uint64_t)i * stride), q);
  for (uint32_t d = FOLD_LL / 2; d >= 1; d >>= 1) {
    if (q == 0) sh[L] = acc;
    __syncthreads();
    if (L < d) acc = g1r_add_quad(acc, sh[L + d], q);
    __syncthreads();
  }
  if (t == 0) {
    st_g1r(rc2 + dst, acc);
    if (dst >= RC_ROWS && dst < RC_ROWS + RC_COLS) st_g1r(rc2 + dst + RC_COLS, G1R::identity());   // the second half-column of the 2^15 layout
  }
}
#endif

#if PLONK_MSM_NB_BITS > 15
int msm_buckets_bits() { return MSM_NB_BITS; }
#endif
// One commitment group through the pipeline: bucket grouping, accumulation, bucket sums, reduction tail.
// bt arrives filled (scalars, sizes, outputs, table); ksl / wide / heavy_thresh are decided here.
// phase: 1 = everything up to the bucket sums of the commitments [bt.kb0, bt.kb0 + bt.count) of the group (bucket sort,
// accumulation, bucket sums, heavy buckets), 2 = the group's reduction tail (row / column sums ... bit sums) over all
// bt.group_count commitments, 3 = both (the ordinary grouped launch).  Round 6: a group whose scalars arrive column by
// column over PCIe (plonk_prover_prove) runs phase 1 per column as it lands and phase 2 once — the throughput-bound
// part overlaps the copies, the latency-bound tail is still paid once per group.
int msm_batch_device_v(Ctx* c, MsmBatch& bt, uint64_t mmax, bool bit_sums, int phase) {
  MsmWork& w = c->msm;
  hipStream_t st = c->stream;
  const int count = bt.count;
  const void* table = bt.table;
  int rc = PLONK_OK;
  bt.ksl = msm_ksl(c, mmax);
  bt.wide = msm_needs_wide_words(bt.rows, bt.table_n) ? 1u : 0u;
  if (MSM_NB_BITS > 15 && !bit_sums) return (plonk::set_last_error("msm", "the 2^19-bucket variant only emits bit sums", __FILE__, __LINE__), PLONK_ERR_ARG);
  {
    const uint64_t avg_slices = (MSM_W * mmax) / bt.ksl / MSM_NB;   // per bucket, uniform digits
    // heavy = well above the expected slice count (skewed digits): twice the uniform average, at least 16 slices
    bt.heavy_thresh = (uint32_t)(2 * avg_slices > 16 ? 2 * avg_slices : 16);
    // (round 6, measured: a threshold of 4 or 8 slices for the many-bucket variant — the listed buckets' chains of up to 15
    // dependent quad additions are the longest of msm_bucket_sum — moves the dense 2^20 proof by -0.1 ms, bench-like by 0 and
    // the widget workload by +0.1: not adopted, profiles/r06b/heavy_ab.jsonl)
  }
  // PLONK_PROF_FINE=1: every phase of the group on its own slot (16 + phase for groups of >= 3 commitments, 24 + phase below;
  // phases: 0 bucket sort, 1 accumulation, 2 bucket sums, 3 heavy buckets, 4 row / column sums (+ fold), 5 bit sums) —
  // hipEvents between the launches, i.e. kernel times WITHOUT a profiler attached (tools/msm_phases.py)
  const bool fine = c->cfg.prof_fine;
  const int fbase = bt.group_count >= 3 ? 16 : 24;
#define FINE_BEGIN(ph) do { if (fine) prof_begin(c, fbase + (ph)); } while (0)
#define FINE_END(ph) do { if (fine) prof_end(c, fbase + (ph)); } while (0)
  const bool tail_quad = !c->cfg.tail_serial;
  if (phase & 1) {
  bt.ordered = (msm_acc_ordered(c, bt.ksl) && !c->cfg.acc_lds) ? 1u : 0u;
  prof_begin(c, 2);
  FINE_BEGIN(0);
  rc = msm_group_sort(c, bt, mmax);
  FINE_END(0);
  prof_end(c, 2);
  if (rc) return rc;
  // upper bound on slices known on the host: no device->host sync on the path
  const uint64_t max_slices = (MSM_W * mmax) / bt.ksl + MSM_NB + 1;
  prof_begin(c, 1);
  FINE_BEGIN(1);
  // PLONK_MSM_ACC=lds: the three-waves-per-SIMD variant (table entries prefetched into LDS) — measured EQUAL to the
  // default on the same box (26.6-26.8 vs 26.8-26.9 ms per proof): the kernel is bound by VALU issue, not by occupancy
  const bool acc_lds = c->cfg.acc_lds;
  // lanes in order of slice length (msm_slices_kernel<true> + msm_accumulate_ordered_kernel): the default from 16-entry slices on
  // (r03a same-box A/B at 2^20: accumulate 26.5 -> 25.7 ms per proof, the waves no longer wait for their longest lane;
  // round 6: also at 16-entry slices, msm_acc_ordered); with 4- / 8-entry slices the accumulation is latency-bound and the
  // ordering gains nothing (r02e: +1.1 ms at 2^16 with the kernels of the time; round 6: +-0).  PLONK_MSM_ORDER=1 / 0 forces either.
  const bool acc_ordered = msm_acc_ordered(c, bt.ksl);
  if (acc_ordered && !acc_lds) {
    rc = msm_order_slices(c, bt);
    if (rc) return rc;
    const uint32_t acc_wg = c->cfg.acc_wg == 64 ? 64u : 128u;   // PLONK_MSM_ACC_WG=64: one wave per workgroup (A/B, round 5)
    hipLaunchKernelGGL(msm_accumulate_ordered_kernel, dim3((uint32_t)((max_slices + acc_wg - 1) / acc_wg), count), dim3(acc_wg), 0, st,
                       (const G1AffineR*)table, bt, w.entries, w.offsets, w.slice_off, w.full_off, w.part_list, (G1RSlot*)w.partial, (G1RSlot*)w.buckets);
  } else if (acc_lds)
    hipLaunchKernelGGL(msm_accumulate_lds_kernel, dim3((uint32_t)((max_slices + 127) / 128), count), dim3(128), 0, st,
                       (const G1AffineR*)table, bt, w.entries, w.offsets, w.slice_off, (G1RSlot*)w.partial);
  else
    hipLaunchKernelGGL(msm_accumulate_kernel, dim3((uint32_t)((max_slices + 127) / 128), count), dim3(128), 0, st,
                       (const G1AffineR*)table, bt, w.entries, w.offsets, w.slice_off, (G1RSlot*)w.partial);
  FINE_END(1);
  prof_end(c, 1);
  if (c->acc_done) HIP_TRY(hipEventRecord(c->acc_done, st));
  prof_begin(c, 2);
  FINE_BEGIN(2);
  {
    const uint64_t avg_slices = (MSM_W * mmax) / bt.ksl / MSM_NB;   // per bucket, uniform digits
    const uint32_t heavy_thresh = bt.heavy_thresh;                  // the list of heavy buckets was written by msm_slices_kernel
    // PLONK_MSM_TAIL=serial: one lane per addition in the heavy-bucket, row/column and bit-sum kernels (A/B, fallback)
    const bool tail_quad_ = !c->cfg.tail_serial;
    const uint32_t fused = (tail_quad_ || MSM_NB_BITS > 15) ? HEAVY_FUSED_WGS : 0u;   // segment workers inside msm_bucket_sum's grid
    const bool list_mode = MSM_NB_BITS > 15 && acc_ordered && !acc_lds;   // bucket sums driven by msm_layout_apply's list
    // few slices per bucket (small MSMs over 2^15 buckets): a quad per bucket (dense_quad above); PLONK_MSM_BSUM=lane restores
    // the one-lane-per-bucket kernels for A/B
    const bool bsum_lane = c->cfg.bsum_lane;
    const bool dense_quad = !list_mode && tail_quad_ && !bsum_lane && avg_slices <= 8;
#define BSUM(G) hipLaunchKernelGGL(msm_bucket_sum_kernel<G>, dim3((list_mode ? 1024u : (dense_quad ? MSM_NB * 4 / 128 : MSM_NB * G / 128)) + fused, count), dim3(128), 0, st, bt, \
                                   (const G1RSlot*)w.partial, w.slice_off, (G1RSlot*)w.buckets, heavy_thresh, w.nheavy, (const HeavyItem*)w.heavy_list, \
                                   (G1RSlot*)w.seg_sum, w.cap_segs, fused, (acc_ordered && !acc_lds) ? 1u : 0u, \
                                   list_mode ? (const uint32_t*)w.multi_list : (const uint32_t*)nullptr, dense_quad ? 1u : 0u)
    if (dense_quad) BSUM(1);
    else if (avg_slices <= 4) BSUM(1);
    else if (avg_slices <= 16) BSUM(2);
    else if (avg_slices <= 32) BSUM(4);
    else BSUM(8);
#undef BSUM
  }
  FINE_END(2);
  FINE_BEGIN(3);
  if (tail_quad || MSM_NB_BITS > 15) {
    hipLaunchKernelGGL(msm_heavy_bucket_quad_kernel, dim3(256, count), dim3(256), 0, st, w.nheavy, (const HeavyItem*)w.heavy_list,
                       (const G1RSlot*)w.seg_sum, w.cap_segs, (G1RSlot*)w.buckets, HEAVY_FUSED_WGS, bt.kb0);
  } else {
    hipLaunchKernelGGL(msm_heavy_seg_kernel, dim3(4 * HEAVY_WGS, count), dim3(64), 0, st, bt, (const G1RSlot*)w.partial, w.slice_off,
                       w.nheavy, (const HeavyItem*)w.heavy_list, (G1RSlot*)w.seg_sum, w.cap_segs);
    hipLaunchKernelGGL(msm_heavy_bucket_kernel, dim3(256, count), dim3(256), 0, st, w.nheavy, (const HeavyItem*)w.heavy_list,
                       (const G1RSlot*)w.seg_sum, w.cap_segs, (G1RSlot*)w.buckets, bt.kb0);
  }
  FINE_END(3);
  if (!(phase & 2)) { prof_end(c, 2); HIP_TRY(hipGetLastError()); return PLONK_OK; }
  } else {
    prof_begin(c, 2);
  }
  // ---- the group's reduction tail: over every commitment of the group
  bt.kb0 = 0;
  bt.count = bt.group_count;
  const int count_g = bt.group_count;
  c->msm.last_rowbits = 8 + (MSM_NB_BITS - 15);
#if PLONK_MSM_NB_BITS > 15
  {   // many buckets: throughput row / column sums, the fold to the 2^15 shapes, then the same bit sums (outputs shifted by E)
    G1RSlot* rc1 = (G1RSlot*)w.chunk;
    G1RSlot* rc2 = rc1 + (size_t)TP_RC1 * MSM_MAX_BATCH;
    // PLONK_MSM_RCTREE=lane: the partial sums of a row / column combined by a lane tree instead of quads (A/B)
    const bool rc_lane_tree = c->cfg.rc_lane_tree;
#define TPK(LPS) do { if (rc_lane_tree) hipLaunchKernelGGL((msm_rowcol_tp_kernel<LPS, false>), dim3(TP_ROWS * LPS / 128 + TP_PARTS * LPS, count_g), dim3(128), 0, st, (const G1RSlot*)w.buckets, rc1); \
                      else hipLaunchKernelGGL((msm_rowcol_tp_kernel<LPS, true>), dim3(TP_ROWS * LPS / 128 + TP_PARTS * LPS, count_g), dim3(128), 0, st, (const G1RSlot*)w.buckets, rc1); } while (0)
    // lanes per sum of 128 buckets: the chip holds 2^17 lanes at two waves per SIMD and a commitment has 2^13 sums; fewer
    // lanes per sum waste less of the tree steps (useful additions per lane-step: LPS 4: 96 %, 8: 88 %, 16: 72 %, 32: 50 %)
    // but lengthen the serial chain.  Measured (profiles/r03e section 7): 8 / 8 / 16 for groups of >= 3 / 2 / 1 commitments
    // (until then 8 / 16 / 32: the small groups over-subscribed the chip two-fold for the sake of depth); 4 for the large
    // groups gains nothing at 2^20 and loses 0.5 ms at 2^22, where the kernel shares the chip with longer transforms.
    const int lps_env = c->cfg.lps;   // PLONK_MSM_LPS: A/B runs
    int lps = count_g >= 2 ? 8 : 16;
    // 2^17 buckets (r04): a quarter of the sums — 8 lanes per sum left half of the chip idle behind 18 dependent additions
    // (0.64 ms per group of four, profiles/r04d); lanes per sum so that a launch has about the chip's 2^17 lane slots
    if (MSM_NB_BITS == 17) lps = count_g >= 3 ? 16 : 32;
    if (lps_env == 4 || lps_env == 8 || lps_env == 16 || lps_env == 32) lps = lps_env;
    else if (lps_env == 1) lps = count_g >= 3 ? 8 : (count_g == 2 ? 16 : 32);   // the old rule
    FINE_BEGIN(4);
    if (lps == 4) TPK(4); else if (lps == 8) TPK(8); else if (lps == 16) TPK(16); else TPK(32);
#undef TPK
    hipLaunchKernelGGL(msm_fold_quad_kernel, dim3(256 + TP_NP + 128, count_g), dim3(64), 0, st, (const G1RSlot*)rc1, rc2);
    FINE_END(4);
    FINE_BEGIN(5);
    hipLaunchKernelGGL(msm_bits_quad_kernel, dim3(17 + TP_E, count_g), dim3(256), 0, st, bt, (const G1RSlot*)rc2, (uint32_t)TP_RC2, (uint32_t)TP_E);
    FINE_END(5);
    prof_end(c, 2);
    HIP_TRY(hipGetLastError());
    return PLONK_OK;
  }
#else
  if (bit_sums && tail_quad) {
    // waves per sum so that the 512 sums per commitment fit the chip's wave slots in one round (1024 SIMDs x 2 waves)
    FINE_BEGIN(4);
    const int rcwv_env = c->cfg.rcwv;   // PLONK_MSM_RCWV, A/B runs: waves per sum for groups of 1 / 2 / >= 3 commitments as a 3-digit number, e.g. 421
    int wv = count_g >= 3 ? 1 : (count_g == 2 ? 2 : 4);
    if (rcwv_env >= 111) wv = count_g >= 3 ? rcwv_env % 10 : (count_g == 2 ? (rcwv_env / 10) % 10 : rcwv_env / 100);
    if (wv == 1) hipLaunchKernelGGL(msm_rowcol_quad_kernel<1>, dim3(RCQ_SUMS, count_g), dim3(64), 0, st, (const G1RSlot*)w.buckets, (G1RSlot*)w.chunk);
    else if (wv == 2) hipLaunchKernelGGL(msm_rowcol_quad_kernel<2>, dim3(RCQ_SUMS, count_g), dim3(128), 0, st, (const G1RSlot*)w.buckets, (G1RSlot*)w.chunk);
    else hipLaunchKernelGGL(msm_rowcol_quad_kernel<4>, dim3(RCQ_SUMS, count_g), dim3(256), 0, st, (const G1RSlot*)w.buckets, (G1RSlot*)w.chunk);
    FINE_END(4);
    FINE_BEGIN(5);
    hipLaunchKernelGGL(msm_bits_quad_kernel, dim3(17, count_g), dim3(256), 0, st, bt, (const G1RSlot*)w.chunk, (uint32_t)RCQ_SUMS, 0u);
    FINE_END(5);
    prof_end(c, 2);
    HIP_TRY(hipGetLastError());
    return PLONK_OK;
  }
  hipLaunchKernelGGL(msm_rowcol_kernel, dim3(RC_WG, count_g), dim3(256), 0, st, (const G1RSlot*)w.buckets,
                     (G1RSlot*)w.chunk);
  if (bit_sums) {
    hipLaunchKernelGGL(msm_bits_kernel, dim3(17, count_g), dim3(128), 0, st, bt, (const G1RSlot*)w.chunk);
  } else {
    constexpr size_t smem = sizeof(G1R) * (RC_ROWS + RC_COLS);
    smem_opt_in(c, (const void*)msm_final_kernel, smem);
    hipLaunchKernelGGL(msm_final_kernel, dim3(count_g), dim3(384), smem, st, bt, (const G1RSlot*)w.chunk);
  }
  prof_end(c, 2);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
#endif
#undef FINE_BEGIN
#undef FINE_END
}


}  // namespace PLONK_MSM_NS

#if PLONK_MSM_NB_BITS == 15   // ---- shared: buffers, variant choice, single-result entry points ----
// upper bound of the slice count over every m <= cap and either bucket count: 2^15 buckets use slices of 4..32 entries
// (MSM_W * m / ksl(m) <= 16 * 2^15 below 2^20), 2^19 buckets slices of 32; + one partial slice per bucket
static uint64_t msm_slice_cap(const Ctx* c, uint64_t cap) {
  if (c->cfg.ksl >= 4) return (uint64_t)MSM_W * cap / c->cfg.ksl + MSM_NB_MAX + 1;
  const uint64_t nb15 = 1ull << 15;
  const uint64_t small = (uint64_t)MSM_W * cap / 4 < (uint64_t)MSM_W * nb15 ? (uint64_t)MSM_W * cap / 4 : (uint64_t)MSM_W * nb15;
  const uint64_t large = (uint64_t)MSM_W * cap / MSM_KSL;
  return (small > large ? small : large) + MSM_NB_MAX + 1;
}

int msm_sort_reserve_fixed(Ctx* c) {   // the bucket sort's size-independent buffers (msm_sort.hip), sized for the larger bucket count
  MsmWork& w = c->msm;
  constexpr int KB = MSM_MAX_BATCH;
  constexpr uint32_t COARSE = 2048;
  HIP_TRY(hipMalloc((void**)&w.coarse_cnt, sizeof(uint32_t) * COARSE * KB));
  HIP_TRY(hipMalloc((void**)&w.coarse_off, sizeof(uint32_t) * (COARSE + 1) * KB));
  HIP_TRY(hipMalloc((void**)&w.coarse_cur, sizeof(uint32_t) * COARSE * KB));
  HIP_TRY(hipMalloc((void**)&w.big_off, sizeof(uint32_t) * (COARSE + 1) * KB));
  HIP_TRY(hipMalloc((void**)&w.big_cnt, sizeof(uint32_t) * 2 * MSM_NB_MAX * KB));   // bin-wide bucket counts, then the run cursors
  HIP_TRY(hipMalloc((void**)&w.full_off, sizeof(uint32_t) * (MSM_NB_MAX + 1) * KB));
  HIP_TRY(hipMalloc((void**)&w.part_list, sizeof(uint32_t) * (MSM_NB_MAX + 1) * KB));
  HIP_TRY(hipMalloc((void**)&w.multi_list, sizeof(uint32_t) * MSM_NB_MAX * KB));
  HIP_TRY(hipMalloc((void**)&w.layout, sizeof(uint32_t) * (2 * (MSM_NB_MAX / 1024) + (MSM_NB_MAX / 1024 + 1) * 132) * KB));
  return PLONK_OK;
}

int msm_reserve(Ctx* c, uint64_t m) {
  MsmWork& w = c->msm;
  constexpr int KB = MSM_MAX_BATCH;
  if (!w.fixed_ok) {
    // all-or-nothing: a failure half way must not leave some of these set (the next call would skip the block and
    // launch kernels on null pointers) — free whatever exists and start over
    for (void** q : {(void**)&w.offsets, (void**)&w.slice_off, (void**)&w.nheavy, &w.heavy_list, (void**)&w.coarse_cnt, (void**)&w.coarse_off,
                     (void**)&w.coarse_cur, (void**)&w.big_off, (void**)&w.big_cnt, (void**)&w.full_off, (void**)&w.part_list, (void**)&w.layout, (void**)&w.multi_list, &w.buckets, &w.chunk,
                     (void**)&w.result}) {
      if (*q) { (void)hipFree(*q); *q = nullptr; }
    }
    if (w.result_host) { (void)hipHostFree(w.result_host); w.result_host = nullptr; }
    HIP_TRY(hipMalloc((void**)&w.offsets, sizeof(uint32_t) * (MSM_NB_MAX + 1) * KB));
    HIP_TRY(hipMalloc((void**)&w.slice_off, sizeof(uint32_t) * (MSM_NB_MAX + 1) * KB));
    HIP_TRY(hipMalloc((void**)&w.nheavy, sizeof(uint32_t) * 4 * KB));
    HIP_TRY(hipMalloc((void**)&w.heavy_list, sizeof(HeavyItem) * MSM_NB_MAX * KB));
    { const int rc_s = msm_sort_reserve_fixed(c); if (rc_s) return rc_s; }
    HIP_TRY(hipMalloc((void**)&w.buckets, sizeof(G1RSlot) * MSM_NB_MAX * KB));
    // row / column sums: 512 per commitment (2^15 buckets); 2 x 4096 stage-1 sums + 528 folded sums (2^19 buckets)
    HIP_TRY(hipMalloc((void**)&w.chunk, sizeof(G1RSlot) * (2 * (MSM_NB_MAX / 128) + 1024) * KB));
    HIP_TRY(hipMalloc((void**)&w.result, sizeof(G1) * MSM_BIT_SUMS * KB));          // one point, or the bit sums of every commitment of a group
    HIP_TRY(hipHostMalloc((void**)&w.result_host, sizeof(G1) * MSM_BIT_SUMS * KB, hipHostMallocDefault));
    w.fixed_ok = true;
  }
  //  the slice capacity depends on the forced slice length too: plonk_ctx_set_config may lower cfg.ksl after the
  // buffers were sized — re-size whenever what this m needs exceeds what `partial` holds, not only when m grows
  if (m > w.cap_m || msm_slice_cap(c, w.cap_m) > w.cap_slices) {
    const uint64_t cap = m > w.cap_m ? m : w.cap_m;
    // release first (the stream may still be reading the old buffers), and forget the old capacity so
    // that a failed reallocation cannot leave a stale cap_m pointing at freed memory
    HIP_TRY(hipStreamSynchronize(c->stream));
    w.cap_m = 0;
    for (void** q : {(void**)&w.tmp_words, (void**)&w.entries, (void**)&w.partial, &w.seg_sum}) {
      if (*q) { HIP_TRY(hipFree(*q)); *q = nullptr; }
    }
    HIP_TRY(hipMalloc((void**)&w.tmp_words, sizeof(uint64_t) * MSM_W * cap * KB));  // words grouped by coarse bin (two planes for tables above 2^27 entries)
    HIP_TRY(hipMalloc((void**)&w.entries, sizeof(uint32_t) * MSM_W * cap * KB));    // entries grouped by bucket
    w.cap_slices = msm_slice_cap(c, cap);
    HIP_TRY(hipMalloc((void**)&w.partial, sizeof(G1RSlot) * w.cap_slices * KB));
    w.cap_segs = w.cap_slices / 128 + MSM_NB_MAX + 1;   // sum over heavy buckets of ceil(slices / HEAVY_SEG)
    HIP_TRY(hipMalloc((void**)&w.seg_sum, sizeof(G1RSlot) * w.cap_segs * KB));
    w.cap_m = cap;
  }
  return PLONK_OK;
}

// The choices msm_batch_device makes for a group, in one place — the launcher follows it, plonk_ctx_describe_msm /
// plonk_ctx_last_msm report it (the variant tests assert it: a switch that is silently ignored fails a test).
// Bucket count by the number of terms: more than 2^18 terms 2^19 buckets, else 2^15.  Round 4 moved the crossover down from
// 2^19 (same box, 2^19 gates: window rows 19.94 ms, bit-position rows over 2^15 / 2^17 / 2^19 buckets 21.31 / 19.99 / 19.20;
// 2^18 gates: 10.94 / 10.92 / 11.01 / 11.44 — profiles/r04h) — a rank of a 2-GPU job at 2^20 gates or of an 8-GPU job at
// 2^22 holds 2^19 points.  plonk_gpu_config.msm_bucket_bits (PLONK_MSM_BUCKETS=15 / 19) forces one wherever it is possible
// (19 needs bit-position or half-density rows and the bit-sum tail; 17: the opt-in A/B build with the 2^17-bucket variant).
void msm_plan(const Ctx* c, uint32_t table_rows, uint64_t table_n, uint64_t mmax, int count, bool bit_sums, plonk_msm_plan_internal* out) {
  (void)count;
  const int buckets_env = c->cfg.bucket_bits;
  const bool can_large = (table_rows == MSM_ROWS_BITPOS || table_rows == MSM_ROWS_HALFPOS) && bit_sums && !c->cfg.tail_serial && !c->cfg.acc_lds;
  int nb_bits = 15;
  if (can_large) {
    if (buckets_env == 17 || buckets_env == 19) nb_bits = buckets_env;
    else if (buckets_env > 15) nb_bits = 19;
    else if (buckets_env != 15) nb_bits = mmax > (1ull << 18) + 64 ? 19 : 15;
  }
#ifndef PLONK_MSM_WITH_MEDIUM
  if (nb_bits == 17) nb_bits = 19;
#endif
  plonk_msm_plan_internal p;
  p.table_rows = table_rows;
  p.bucket_bits = (uint32_t)nb_bits;
  p.digit_width = table_rows == MSM_ROWS_WINDOW ? 16u : (table_rows == MSM_ROWS_BITPOS ? (uint32_t)nb_bits + 2u : (uint32_t)nb_bits + 1u);
  p.terms = mmax;
  bool ordered;
  if (nb_bits == 19) { p.slice_entries = nbl::msm_ksl(c, mmax); ordered = nbl::msm_acc_ordered(c, p.slice_entries); p.wide_words = nbl::msm_needs_wide_words(table_rows, table_n); }
#ifdef PLONK_MSM_WITH_MEDIUM
  else if (nb_bits == 17) { p.slice_entries = nbm::msm_ksl(c, mmax); ordered = nbm::msm_acc_ordered(c, p.slice_entries); p.wide_words = nbm::msm_needs_wide_words(table_rows, table_n); }
#endif
  else { p.slice_entries = nb15::msm_ksl(c, mmax); ordered = nb15::msm_acc_ordered(c, p.slice_entries); p.wide_words = nb15::msm_needs_wide_words(table_rows, table_n); }
  p.ordered_lanes = ordered && !c->cfg.acc_lds;
  static const char* const names[3][3] = {
      {"nb15::msm_accumulate_kernel", "nb15::msm_accumulate_ordered_kernel", "nb15::msm_accumulate_lds_kernel"},
      {"nbm::msm_accumulate_kernel", "nbm::msm_accumulate_ordered_kernel", "nbm::msm_accumulate_lds_kernel"},
      {"nbl::msm_accumulate_kernel", "nbl::msm_accumulate_ordered_kernel", "nbl::msm_accumulate_lds_kernel"}};
  const uint64_t avg_slices = (MSM_W * mmax) / (p.slice_entries ? p.slice_entries : 1) / (1ull << nb_bits);
  p.flags = ((nb_bits > 15 && c->cfg.sort13 == 1) ? PLONK_PLAN_SORT13 : 0u) | (c->cfg.tail_serial ? PLONK_PLAN_TAIL_SERIAL : 0u) | (c->cfg.acc_lds ? PLONK_PLAN_ACCUMULATE_LDS : 0u) |
            ((c->cfg.bsum_lane || c->cfg.tail_serial || avg_slices > 8 || (nb_bits > 15 && p.ordered_lanes)) ? PLONK_PLAN_BUCKET_SUM_LANE : 0u);
  p.kernel = names[nb_bits == 19 ? 2 : (nb_bits == 17 ? 1 : 0)][c->cfg.acc_lds ? 2 : (p.ordered_lanes ? 1 : 0)];
  *out = p;
}

// `count` (<= MSM_MAX_BATCH) independent MSMs over the same bases, launched together: the
// latency-bound reduction kernels run once per group instead of once per commitment
// (Prover::commit_polynomials' 4-way fan-out, prover.rs:187-210).  m[k] == 0 -> identity.
// Which bucket count: 2^19 buckets (nb19) for bit-position tables, the bit-sum tail and more than 2^19 terms — 12.1
// instead of 14.7 additions per scalar, a bucket is one lane of ~24 entries — else 2^15 (nb15).  PLONK_MSM_BUCKETS=15 / 19
// forces either wherever it is possible (19 needs bit-position tables and the bit-sum tail).
int msm_batch_device(Ctx* c, const Fr* const* scalars_dev, const uint64_t* m, int count, G1* const* out_dev, bool bit_sums,
                     const void* table, uint64_t table_n, const Fr* const* tail_dev, const uint64_t* split, uint32_t table_rows,
                     int phase, int kb0, int kcount) {
  if (count <= 0) return PLONK_OK;
  if (count > MSM_MAX_BATCH) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  if (kcount < 0) kcount = count - kb0;
  if (phase < 1 || phase > 3 || kb0 < 0 || kcount < 1 || kb0 + kcount > count || (phase == 3 && (kb0 != 0 || kcount != count)))
    return (plonk::set_last_error("invalid argument", "msm_batch_device: phase / column range", __FILE__, __LINE__), PLONK_ERR_ARG);
  if (!table) { table = c->srs_table; table_n = c->srs_n; table_rows = c->srs_rows; }
  if (table && table_rows != MSM_ROWS_WINDOW && table_rows != MSM_ROWS_BITPOS && table_rows != MSM_ROWS_HALFPOS) return (plonk::set_last_error("invalid argument", "msm: table rows", __FILE__, __LINE__), PLONK_ERR_ARG);
  uint64_t mmax = 0;
  for (int k = 0; k < count; ++k) {
    if (m[k] > table_n) return PLONK_ERR_DEGREE;
    if (m[k] > mmax) mmax = m[k];
  }
  hipStream_t st = c->stream;
  if (mmax == 0) {
    if (!(phase & 2)) return PLONK_OK;   // (a group in parts: the identities are written with the tail)
    c->msm.last_rowbits = 8;
    for (int k = 0; k < count; ++k)
      for (int j = 0; j < (bit_sums ? MSM_BIT_SUMS : 1); ++j) hipLaunchKernelGGL(msm_identity_kernel, dim3(1), dim3(64), 0, st, out_dev[k] + j);
    HIP_TRY(hipGetLastError());
    if (c->acc_done) HIP_TRY(hipEventRecord(c->acc_done, st));   // callers gate side-stream work on it: never leave a stale event
    return PLONK_OK;
  }
  if (!table) return PLONK_ERR_NO_SRS;
  int rc = msm_reserve(c, mmax);
  if (rc) return rc;
  MsmWork& w = c->msm;
  MsmBatch bt{};
  bt.count = (phase & 1) ? kcount : count;
  bt.kb0 = (phase & 1) ? kb0 : 0;
  bt.group_count = count;
  bt.cap_m = w.cap_m;
  bt.cap_slices = w.cap_slices;
  bt.table = table;
  bt.table_n = table_n;
  bt.rows = table_rows;
  for (int k = 0; k < count; ++k) {
    bt.scalars[k] = scalars_dev[k]; bt.m[k] = m[k]; bt.out[k] = out_dev[k];
    bt.tail[k] = tail_dev ? tail_dev[k] : nullptr;
    bt.split[k] = (tail_dev && split) ? split[k] : ~0ull;
  }
  plonk_msm_plan_internal plan;
  msm_plan(c, table_rows, table_n, mmax, count, bit_sums, &plan);
  c->last_plan = plan;
  const int nb_bits = (int)plan.bucket_bits;
  if (nb_bits == 19) return nbl::msm_batch_device_v(c, bt, mmax, bit_sums, phase);
#ifdef PLONK_MSM_WITH_MEDIUM
  if (nb_bits == 17) return nbm::msm_batch_device_v(c, bt, mmax, bit_sums, phase);
#else
  if (nb_bits == 17) return nbl::msm_batch_device_v(c, bt, mmax, bit_sums, phase);
#endif
  return nb15::msm_batch_device_v(c, bt, mmax, bit_sums, phase);
}

int msm_device(Ctx* c, const Fr* scalars_dev, uint64_t m, G1* out_dev) {
  return msm_batch_device(c, &scalars_dev, &m, 1, &out_dev, false);
}

int xyzz_to_affine97_device(Ctx* c, const G1* in_dev, uint8_t* out97_dev) {
  hipLaunchKernelGGL(xyzz_to_affine97_kernel, dim3(1), dim3(64), 0, c->stream, in_dev, out97_dev);
  HIP_TRY(hipGetLastError());
  return PLONK_OK;
}
#endif   // shared

}  // namespace plonk
