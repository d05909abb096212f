// Random 128-byte gathers over tables of growing size: does the window-table gather of msm_accumulate keep its rate
// when the tables grow from 2 GiB (16 window positions) to 32+ GiB (one table per bit position, round 3)?
// Each lane reads whole 128-B entries (8 x dwordx4... issued as 2 x 4) at hashed indices, G gathers in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

template <int G>
__global__ void __launch_bounds__(256) gather(const uint4* __restrict__ tab, uint64_t entries, int iters, uint32_t* out, int lds_pad) {
  extern __shared__ uint32_t pad[];
  uint64_t s = mix(blockIdx.x * 256ull + threadIdx.x + 1);
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint4 v[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      s = mix(s + g + 1);
      const uint4* p = tab + (s % entries) * 8;
      v[g][0] = p[0]; v[g][1] = p[7];          // first and last 16 B of the entry: the whole 128-B line is fetched
    }
#pragma unroll
    for (int g = 0; g < G; ++g) acc += v[g][0].x ^ v[g][1].w;
  }
  if (lds_pad < 0) pad[threadIdx.x] = acc;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  double sizes_gib[] = {2, 8, 16, 34, 68, 137};
  uint32_t* out; hipMalloc(&out, 4 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (double gib : sizes_gib) {
    size_t bytes = (size_t)(gib * (1ull << 30));
    size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
    if (bytes + (4ull << 30) > fr) { printf("skip %.0f GiB (free %.1f GiB)\n", gib, fr / 1073741824.0); continue; }
    uint4* tab = nullptr;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("hipMalloc %.0f GiB failed\n", gib); continue; }
    hipMemset(tab, 1, bytes); hipDeviceSynchronize();
    uint64_t entries = bytes / 128;
    for (int wps : {2, 4, 8}) {      // waves per SIMD (occupancy limited through dynamic LDS: 160 KB / (wps) per workgroup)
      int lds = wps == 8 ? 0 : (wps == 4 ? 38 << 10 : 78 << 10);
      int blocks = 256 * wps * 4, iters = 64;
      auto launch = [&](int g) {
        if (g == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), lds, 0, tab, entries, iters * 2, out, lds);
        else hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), lds, 0, tab, entries, iters, out, lds);
      };
      for (int g : {1, 2}) {
        if (lds) { hipFuncSetAttribute((const void*)gather<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                   hipFuncSetAttribute((const void*)gather<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
        launch(g); hipDeviceSynchronize();
        hipEventRecord(e0); launch(g); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double gathers = (double)blocks * 256 * iters * 2;
        printf("table %6.1f GiB  waves/SIMD=%d in-flight=%d  %8.3f ms  %7.2f G gathers/s  %7.1f GB/s (128 B lines)\n", gib, wps, g, ms,
               gathers / ms / 1e6, gathers * 128 / ms / 1e6);
      }
    }
    hipFree(tab);
  }
  return 0;
}
