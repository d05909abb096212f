// Measured prototype: shared-inversion AFFINE additions against the XYZZ mixed addition that
// msm_accumulate runs today, on the product's own field / curve code (fp28.cuh, curve28.cuh, fp_safegcd.cuh).
//
//   xyzz_chain     lane: acc += T[idx]  K times (G1R::add_affine: 8 products + 2 squarings, 9 reductions), table entry of
//                  step k + 1 prefetched as in msm_accumulate_kernel.
//   affine_batch   lane: K INDEPENDENT sums P_k + Q_k of table points with ONE inversion (Montgomery's trick along the
//                  lane's own K pairs): pass 1 forms dx_k = xQ - xP and the running products (parked in a global scratch
//                  array, [k][lane] layout), the total is inverted with the constant-time safegcd inverse (~23 k
//                  instructions instead of ~300 k for Fermat), pass 2 walks back: 1/dx_k, lambda, x3, y3 — 5 products
//                  + 1 squaring per sum + (inversion / K).  This is the arithmetic a pairwise-tree accumulation over
//                  bucket entries would run; its extra traffic (second gather of the operands, running products out and
//                  back, affine results out) is part of the measurement.
// Output: ns per addition per lane-slot for both, and a cross-check of the affine sums against the XYZZ addition.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../plonk_amd/csrc/curve28.cuh"
#include "../../plonk_amd/csrc/fp_safegcd.cuh"
using namespace plonk;

__device__ __forceinline__ Fp28 ld_f28(const Fp28Slot* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  Fp28 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w; r.l[12] = d.x; r.l[13] = d.y;
  return r;
}
__device__ __forceinline__ void st_f28(Fp28Slot* p, const Fp28& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]); q[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// table[i] = hash(i) * G, affine (one-off)
__global__ void make_table(G1AffineR* table, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t gx[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u, 0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
  const uint32_t gy[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u, 0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
  Fp x, y;
  for (int k = 0; k < 12; ++k) { x.l[k] = gx[k]; y.l[k] = gy[k]; }
  const G1R g = G1R::from_affine(Fp28::from_fp(x), Fp28::from_fp(y));
  const G1R p = g.mul_u32(hash32(i) | 1u);
  Fp28 ax, ay;
  g1r_to_affine(p, &ax, &ay);
  st_f28(&table[i].x, ax.canon());
  st_f28(&table[i].y, ay.canon());
}

template <int K>
__global__ void __launch_bounds__(128) xyzz_chain(const G1AffineR* __restrict__ table, uint32_t mask, Fp28Slot* __restrict__ out) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  G1R acc = G1R::identity();
  uint32_t idx = hash32(lane * 977u) & mask;
  Fp28 x = ld_f28(&table[idx].x), y = ld_f28(&table[idx].y);
  for (int k = 0; k < K; ++k) {
    const Fp28 xc = x, yc = y;
    if (k + 1 < K) {
      idx = hash32(idx + k + 1) & mask;
      x = ld_f28(&table[idx].x);
      y = ld_f28(&table[idx].y);
    }
    acc = acc.add_affine(xc, yc);
  }
  st_f28(out + 4ull * lane, acc.X); st_f28(out + 4ull * lane + 1, acc.Y); st_f28(out + 4ull * lane + 2, acc.ZZ); st_f28(out + 4ull * lane + 3, acc.ZZZ);
}

// pairs (P_k, Q_k) = (table[i_k], table[j_k]); results to res[k][lane] (x, y); scratch[k][lane] = running products
template <int K, bool FERMAT>
__global__ void __launch_bounds__(128) affine_batch(const G1AffineR* __restrict__ table, uint32_t mask, Fp28Slot* __restrict__ scratch,
                                                    G1AffineR* __restrict__ res, uint32_t lanes) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  Fp28 run = Fp28::one();
  uint32_t seed = hash32(lane * 31337u + 7u);
  for (int k = 0; k < K; ++k) {
    const uint32_t i = hash32(seed + 2 * k) & mask, j = hash32(seed + 2 * k + 1) & mask;
    const Fp28 dx = Fp28::sub_lazy<4>(ld_f28(&table[j].x), ld_f28(&table[i].x));
    st_f28(scratch + (uint64_t)k * lanes + lane, run);          // product of dx_0 .. dx_{k-1}
    run = Fp28::mul(run, dx);
  }
  Fp28 inv = FERMAT ? fp28_inv(run) : fp28_inv_gcd(run);
  for (int k = K - 1; k >= 0; --k) {
    const uint32_t i = hash32(seed + 2 * k) & mask, j = hash32(seed + 2 * k + 1) & mask;
    const Fp28 x1 = ld_f28(&table[i].x), y1 = ld_f28(&table[i].y), x2 = ld_f28(&table[j].x), y2 = ld_f28(&table[j].y);
    const Fp28 dx = Fp28::sub_lazy<4>(x2, x1);
    const Fp28 inv_k = Fp28::mul(inv, ld_f28(scratch + (uint64_t)k * lanes + lane));   // 1 / dx_k
    inv = Fp28::mul(inv, dx);
    const Fp28 lam = Fp28::mul(Fp28::sub_lazy<4>(y2, y1), inv_k);
    const Fp28 x3 = Fp28::sub<4>(Fp28::sub_lazy<4>(lam.sqr(), x1), x2);                  // < 10p
    const Fp28 y3 = Fp28::sub<4>(Fp28::mul(lam, Fp28::sub_lazy<32>(x1, x3)), y1);        // < 6p
    st_f28(&res[(uint64_t)k * lanes + lane].x, x3);
    st_f28(&res[(uint64_t)k * lanes + lane].y, y3);
  }
}

// cross-check: recompute pair k of lane with the XYZZ addition and compare affine coordinates
template <int K>
__global__ void check(const G1AffineR* table, uint32_t mask, const G1AffineR* res, uint32_t lanes, uint32_t* bad) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= lanes) return;
  const uint32_t seed = hash32(lane * 31337u + 7u);
  const int k = lane % K;
  const uint32_t i = hash32(seed + 2 * k) & mask, j = hash32(seed + 2 * k + 1) & mask;
  if (i == j) return;   // the prototype does not handle P = Q (dx = 0); the product would take its doubling branch
  const G1R s = G1R::from_affine(ld_f28(&table[i].x), ld_f28(&table[i].y)).add_affine(ld_f28(&table[j].x), ld_f28(&table[j].y));
  Fp28 ax, ay;
  g1r_to_affine(s, &ax, &ay);
  const bool ok = ax.eq_mod(ld_f28(&res[(uint64_t)k * lanes + lane].x)) && ay.eq_mod(ld_f28(&res[(uint64_t)k * lanes + lane].y));
  if (!ok) atomicAdd(bad, 1u);
}

template <int K>
static void run_case(const G1AffineR* table, uint32_t mask) {
  const uint32_t lanes = 256 * 4 * 64 * 2;     // xyzz: 2 waves per SIMD over the whole chip (212 VGPRs), as msm_accumulate
  const uint32_t lanes_a = 256 * 4 * 64 * 3;   // affine: 3 waves per SIMD (154 VGPRs)
  Fp28Slot *out, *scratch; G1AffineR* res; uint32_t* bad;
  hipMalloc(&out, sizeof(Fp28Slot) * 4ull * lanes); hipMalloc(&scratch, sizeof(Fp28Slot) * (uint64_t)K * lanes_a);
  hipMalloc(&res, sizeof(G1AffineR) * (uint64_t)K * lanes_a); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms_x = 0, ms_a = 0, ms_a2 = 0, ms_f = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(xyzz_chain<K>, dim3(lanes / 128), dim3(128), 0, 0, table, mask, out); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_x, e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL((affine_batch<K, false>), dim3(lanes / 128), dim3(128), 0, 0, table, mask, scratch, res, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_a2, e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL((affine_batch<K, false>), dim3(lanes_a / 128), dim3(128), 0, 0, table, mask, scratch, res, lanes_a); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_a, e0, e1);
  }
  hipLaunchKernelGGL(check<K>, dim3(lanes_a / 128), dim3(128), 0, 0, table, mask, (const G1AffineR*)res, lanes_a, bad);
  uint32_t hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  if (K <= 32) {
    hipEventRecord(e0); hipLaunchKernelGGL((affine_batch<K, true>), dim3(lanes_a / 128), dim3(128), 0, 0, table, mask, scratch, res, lanes_a); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_f, e0, e1);
  }
  const double px = ms_x * 1e6 / ((double)K * lanes), pa2 = ms_a2 * 1e6 / ((double)K * lanes), pa = ms_a * 1e6 / ((double)K * lanes_a);
  printf("K=%3d table 2^%-2d entries | chip-level ns per addition: xyzz mixed add (2 waves/SIMD) %.4f | affine shared inversion, safegcd: 2 waves/SIMD %.4f, "
         "3 waves/SIMD %.4f = %.2fx the xyzz cost", K, 32 - __builtin_clz(mask), px, pa2, pa, pa / px);
  if (K <= 32) printf(" | with a Fermat inversion %.4f", ms_f * 1e6 / ((double)K * lanes_a));
  printf(" | mismatches %u\n", hb);
  hipFree(out); hipFree(scratch); hipFree(res); hipFree(bad);
}

int main() {
  for (uint32_t logt : {16u, 22u}) {      // 8 MiB (L2-resident) and 512 MiB (HBM gathers) of table
    const uint32_t n = 1u << logt;
    G1AffineR* table; hipMalloc(&table, sizeof(G1AffineR) * (uint64_t)n);
    hipLaunchKernelGGL(make_table, dim3((n + 63) / 64), dim3(64), 0, 0, table, n);
    hipDeviceSynchronize();
    run_case<16>(table, n - 1);
    run_case<32>(table, n - 1);
    run_case<64>(table, n - 1);
    run_case<128>(table, n - 1);
    hipFree(table);
  }
  return 0;
}
