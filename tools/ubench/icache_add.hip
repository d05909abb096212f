// What does an XYZZ point addition cost a LONE wave when its code is not in the instruction cache?  The reduction-tail kernels
// of the MSM (row / column sums, bit sums, fold, bucket sums of small groups) run 8-24 dependent additions per wave from
// two or three inlined copies of the addition (a serial loop + an LDS tree: ~60 KB per single-lane copy, ~22 KB per quad
// copy) on CUs whose 64 KB instruction cache was just flushed by the accumulation kernel — each copy is fetched from L2 / HBM
// and then executed a handful of times.  This measures, with the shader clock inside the kernel (s_memtime):
//   rolled     K dependent additions through ONE copy of G1R::add in a loop   (first trip cold, the rest from the I-cache)
//   unrolled4  the same K additions through FOUR copies (a 4x unrolled loop, ~240 KB: never resident in the I-cache)
// per wave on an otherwise idle chip, for K = 1 .. 64, on the FIRST launch of the process (L2 cold too) and on later launches.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/icache_add.hip -o /tmp/icache_add
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../plonk_amd/csrc/curve28.cuh"
using namespace plonk;

__global__ void make_points(G1R* pts, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t gx[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u, 0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
  const uint32_t gy[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u, 0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
  Fp x, y;
  for (int k = 0; k < 12; ++k) { x.l[k] = gx[k]; y.l[k] = gy[k]; }
  pts[i] = G1R::from_affine(Fp28::from_fp(x), Fp28::from_fp(y)).mul_u32(2u * (uint32_t)i + 3u);
}

template <int UNROLL>
__global__ void __launch_bounds__(64) add_chain(const G1R* __restrict__ pts, int K, uint64_t* __restrict__ cycles, uint32_t* __restrict__ sink) {
  G1R acc = pts[threadIdx.x & 3];
  const long long t0 = clock64();
#pragma unroll UNROLL
  for (int k = 0; k < K; ++k) acc = acc.add(pts[4 + ((k + threadIdx.x) & 63)]);
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = (uint64_t)(t1 - t0);
  if (acc.X.l[0] == 0xdeadbeefu) sink[0] = 1;
}

int main() {
  G1R* pts;
  uint64_t* cyc;
  uint32_t* sink;
  hipMalloc(&pts, sizeof(G1R) * 128);
  hipMalloc(&cyc, 8 * 64);
  hipMalloc(&sink, 4);
  hipLaunchKernelGGL(make_points, dim3(2), dim3(64), 0, 0, pts, 128);
  hipDeviceSynchronize();
  const int Ks[] = {1, 2, 4, 8, 16, 64};
  for (int round = 0; round < 3; ++round) {
    for (int K : Ks) {
      uint64_t c1 = 0, c4 = 0;
      hipLaunchKernelGGL(add_chain<1>, dim3(1), dim3(64), 0, 0, pts, K, cyc, sink);
      hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(add_chain<4>, dim3(1), dim3(64), 0, 0, pts, K, cyc, sink);
      hipMemcpy(&c4, cyc, 8, hipMemcpyDeviceToHost);
      printf("launch round %d  K=%2d additions: rolled %8llu clocks (%7.0f per addition) | unrolled x4 %8llu clocks (%7.0f per addition)\n", round, K,
             (unsigned long long)c1, (double)c1 / K, (unsigned long long)c4, (double)c4 / K);
    }
  }
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("clock64 counts s_memtime ticks; device clock rate attribute %d kHz, wall clock rate %d kHz\n", clk, rate);
  // the same chain timed from the host for K = 64 (rolled), to calibrate the tick
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(add_chain<1>, dim3(1), dim3(64), 0, 0, pts, 4096, cyc, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  uint64_t c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("K=4096 rolled: %.3f ms by hipEvents, %llu clocks in the kernel => %.1f clocks per us; %.2f us per warm addition\n", ms, (unsigned long long)c, c / (ms * 1e3), ms * 1e3 / 4096);
  return 0;
}
