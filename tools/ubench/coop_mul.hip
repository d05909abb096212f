// Latency of ONE Montgomery product over BLS12-381 Fp on a lone wave: the 490 dependent instructions of Fp28::mul on a single
// lane (what every level of a quad addition in the MSM's reduction tail waits for) against the 16-lane cooperative product
// of coop_mul.hpp (one DPP row per product).  DESIGN.md section 7 item 2 sizes the second at about a quarter of the first;
// this measures it.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/coop_mul.hip -o /tmp/coop_mul
//
//   lane_chain   x <- Fp28::mul(A, x), K times, every lane its own chain (lane 0 of wave 0 is what is compared)
//   coop_chain   x <- coop::mul(A, x), K times, every 16-lane row one chain (A uniform: kernel arguments, i.e. SGPRs)
// Both chains start from the same x and must agree modulo p after K steps (checked on the host through Fp28::canon).
// Timed with 1 wave (a lone wave on the chip: the latency that bounds the tail kernels) and with 1, 2, 4 waves per SIMD on
// every CU (the throughput price: a cooperative product occupies 16 lanes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../plonk_amd/csrc/fp28.cuh"
#include "coop_mul.hpp"
using namespace plonk;

struct DppLanes {   // a value per lane; shifts inside the 16-lane DPP row, zero fill (row_shr / row_shl with bound_ctrl)
  using u32 = uint32_t;
  using u64 = uint64_t;
  static __device__ __forceinline__ u64 zero64() { return 0; }
  static __device__ __forceinline__ u64 mad(u64 acc, uint32_t uni, u32 lane) { return acc + (uint64_t)uni * lane; }
  template <int S> static __device__ __forceinline__ u32 shr32(u32 x) {
    if constexpr (S == 0) return x;
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + S, 0xf, 0xf, true);
  }
  template <int S> static __device__ __forceinline__ u32 shl32(u32 x) {
    if constexpr (S == 0) return x;
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + S, 0xf, 0xf, true);
  }
  template <int S> static __device__ __forceinline__ u64 shr64(u64 x) { return (u64)shr32<S>((u32)x) | ((u64)shr32<S>((u32)(x >> 32)) << 32); }
  template <int S> static __device__ __forceinline__ u64 shl64(u64 x) { return (u64)shl32<S>((u32)x) | ((u64)shl32<S>((u32)(x >> 32)) << 32); }
  static __device__ __forceinline__ u32 lane_lt(int n) { return (int)(threadIdx.x & 15) < n ? 0xffffffffu : 0u; }
  static __device__ __forceinline__ u32 lane_eq(int n) { return (int)(threadIdx.x & 15) == n ? 0xffffffffu : 0u; }
  static __device__ __forceinline__ u64 select64(u32 m, u64 a, u64 b) { return m ? a : b; }
  static __device__ __forceinline__ u32 and32(u32 a, u32 b) { return a & b; }
  static __device__ __forceinline__ u32 add32(u32 a, u32 b) { return a + b; }
  static __device__ __forceinline__ u64 add64(u64 a, u64 b) { return a + b; }
  static __device__ __forceinline__ u32 lo32(u64 a) { return (u32)a; }
  static __device__ __forceinline__ u64 widen(u32 a) { return a; }
  static __device__ __forceinline__ u64 shr64_bits(u64 a, int s) { return a >> s; }
  static __device__ __forceinline__ u32 nonzero32(u32 a) { return a ? 0xffffffffu : 0u; }
};

struct Limbs14 { uint32_t l[14]; };

__global__ void __launch_bounds__(256) lane_chain(Limbs14 a, Limbs14 x0, int iters, uint32_t* __restrict__ out) {
  Fp28 A, x;
  for (int i = 0; i < 14; ++i) { A.l[i] = a.l[i]; x.l[i] = x0.l[i] + (i == 1 ? threadIdx.x : 0u); }   // lane 0 runs the reference chain; the others differ, so nothing is scalarised
  for (int k = 0; k < iters; ++k) x = Fp28::mul(A, x);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 14; ++i) out[i] = x.l[i];
  else if (x.l[0] == 0xdeadbeefu) out[14] = 1;   // keep every lane's chain alive
}

__global__ void __launch_bounds__(256) coop_chain(coop::Uniform a, Limbs14 x0, int iters, uint32_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 15;
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) x = lane == (uint32_t)i ? x0.l[i] : x;   // lane j <- limb j, lanes 14 / 15 zero
  for (int k = 0; k < iters; ++k) x = coop::Mul<DppLanes>::mul(a, x);
  if (blockIdx.x == 0 && threadIdx.x < 16) out[threadIdx.x] = x;
  else if (x == 0xdeadbeefu) out[16] = 1;
}

template <class K, class... Args>
static float timed(K kern, int blocks, int threads, Args... args) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, args...);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, args...);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  // A = a fixed residue in Montgomery form (any value below p will do), x0 = another
  Limbs14 a, x0;
  coop::Uniform au;
  uint64_t s = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < 14; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull; a.l[i] = (uint32_t)(s >> 36) & (i == 13 ? 0xffffu : Fp28::MASK);
    s = s * 6364136223846793005ull + 1442695040888963407ull; x0.l[i] = (uint32_t)(s >> 36) & (i == 13 ? 0xffffu : Fp28::MASK);
    au.l[i] = a.l[i];
  }
  uint32_t* out;
  hipMalloc(&out, 256);
  uint32_t h_lane[14], h_coop[16];
  const int K = 4096;
  // ---- correctness: the two chains agree modulo p (after 1, 2, 7 and K steps)
  bool ok = true;
  for (int iters : {1, 2, 7, K}) {
    hipLaunchKernelGGL(lane_chain, dim3(1), dim3(64), 0, 0, a, x0, iters, out);
    hipMemcpy(h_lane, out, sizeof h_lane, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(coop_chain, dim3(1), dim3(64), 0, 0, au, x0, iters, out);
    hipMemcpy(h_coop, out, sizeof h_coop, hipMemcpyDeviceToHost);
    Fp28 L, C;
    for (int i = 0; i < 14; ++i) { L.l[i] = h_lane[i]; C.l[i] = h_coop[i]; }
    C.normalize();
    const Fp28 lc = L.canon(), cc = C.canon();
    bool same = h_coop[14] == 0 && h_coop[15] == 0;
    for (int i = 0; i < 14; ++i) same = same && lc.l[i] == cc.l[i];
    printf("%5d steps: cooperative product %s the single-lane product modulo p\n", iters, same ? "EQUALS" : "DIFFERS FROM");
    ok = ok && same;
  }
  // ---- latency: one wave on the whole chip
  const float t_lane = timed(lane_chain, 1, 64, a, x0, K, out), t_coop = timed(coop_chain, 1, 64, au, x0, K, out);
  printf("lone wave, %d dependent products: single lane %.3f ms = %.0f cycles per product | 16-lane cooperative %.3f ms = %.0f cycles per product | x%.2f\n",
         K, t_lane, t_lane * 1e-3 * 2.4e9 / K, t_coop, t_coop * 1e-3 * 2.4e9 / K, t_lane / t_coop);
  // ---- throughput: every SIMD busy with w waves (256 CUs x 4 SIMDs)
  for (int w : {1, 2, 4}) {
    const int blocks = 256 * w;   // 256 threads = one wave per SIMD of a CU
    const float a_ms = timed(lane_chain, blocks, 256, a, x0, K / 4, out), c_ms = timed(coop_chain, blocks, 256, au, x0, K / 4, out);
    const double prod_lane = (double)blocks * 256 * (K / 4), prod_coop = (double)blocks * 16 * (K / 4);
    printf("waves/SIMD=%d: single lane %.3f ms (%.2f G products/s) | cooperative %.3f ms (%.3f G products/s)\n", w, a_ms, prod_lane / a_ms / 1e6, c_ms,
           prod_coop / c_ms / 1e6);
  }
  hipFree(out);
  return ok ? 0 : 1;
}
