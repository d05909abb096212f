// Does the matrix pipe issue beside the integer VALU on gfx950?  (the one untested idea on the Montgomery
// product — move the CONSTANT half, m * p, onto v_mfma_i32_*_i8 as a Toeplitz product by a fixed byte matrix.)
//
// Kernel A: 64 dependent-chain v_mad_u64_u32 per iteration (8 accumulators, the mix of fp28.cuh's product).
// Kernel B: the same + one v_mfma_i32_32x32x16_i8 after every 8th mad (8 per iteration — the ratio the scheme would need:
//           a 14 x 14-limb product keeps ~260 VALU instructions and hands 49 x 98 byte products = 32 MFMAs per wave to the
//           matrix pipe).
// Kernel C: the 8 MFMAs alone.
// If B takes as long as A, the matrix pipe is free beside the VALU and the scheme's cost is only its layout work
// (limbs -> signed bytes -> operand layout across lane halves, 98 int32 columns back into 28-bit limbs: ~400 VALU
// instructions per product against the 196 + ~60 it removes — see DESIGN.md).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
typedef int v16i __attribute__((ext_vector_type(16)));
#define MAD8 "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
template <int MODE>   // 0: mads, 1: mads + MFMAs, 2: MFMAs
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1;
  uint64_t c0 = a0, c1 = a1, c2 = a0 + 5, c3 = a1 + 7, c4 = a0 + 9, c5 = a1 + 11, c6 = a0 + 13, c7 = a1 + 17;
  v16i acc = {0};
  long ma = 0x0102030405060708L + threadIdx.x, mb = 0x0807060504030201L;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE != 2) asm volatile(MAD8 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(a1) : "vcc");
      if (MODE != 0) asm volatile("v_mfma_i32_32x32x16_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(ma), "v"(mb));
    }
  }
  uint32_t s = (uint32_t)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7);
  for (int j = 0; j < 16; ++j) s += (uint32_t)acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(int waves_per_simd) {
  uint32_t* out; hipMalloc(&out, 4 << 20);
  const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one wave per SIMD of a CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1u); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}
int main() {
  for (int w : {1, 2, 4}) {
    const float a = run<0>(w), b = run<1>(w), c = run<2>(w);
    const double mads = (double)ITER * 64 * w, mf = (double)ITER * 8 * w;
    printf("waves/SIMD=%d  mads only %.3f ms (%.2f cycles per mad per SIMD)   mads + MFMAs %.3f ms (x%.2f)   MFMAs only %.3f ms (%.1f cycles per MFMA per SIMD)\n",
           w, a, a * 1e-3 * 2.4e9 / mads, b, b / a, c, c * 1e-3 * 2.4e9 / mf);
  }
  return 0;
}
