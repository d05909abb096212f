// VALU issue-rate microbenchmark for the integer/f64 instructions a big-int multiplier can be built from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x9e3779b9;
  uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a0 + 9, c5 = a1 + 11, c6 = a2 + 13, c7 = a3 + 17;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = 1.5, d5 = 2.5, d6 = 3.5, d7 = 4.5;
  for (int i = 0; i < ITER; ++i) {
    if (OP == 0) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(a1) : "vcc");) }
    if (OP == 1) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(*(uint32_t*)&c4), "+v"(*(uint32_t*)&c5), "+v"(*(uint32_t*)&c6), "+v"(*(uint32_t*)&c7) : "v"(seed));) }
    if (OP == 2) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(*(uint32_t*)&c4), "+v"(*(uint32_t*)&c5), "+v"(*(uint32_t*)&c6), "+v"(*(uint32_t*)&c7) : "v"(seed));) }
    if (OP == 3) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %0\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(*(uint32_t*)&c4), "+v"(*(uint32_t*)&c5), "+v"(*(uint32_t*)&c6), "+v"(*(uint32_t*)&c7) : "v"(seed));) }
    if (OP == 4) { REP8(asm volatile("v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(*(uint32_t*)&c4), "+v"(*(uint32_t*)&c5), "+v"(*(uint32_t*)&c6), "+v"(*(uint32_t*)&c7) : "v"(seed));) }
    if (OP == 5) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %4\n v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));) }
    if (OP == 6) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %0, vcc\n v_add_co_u32 %4, vcc, %4, %5\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_addc_co_u32 %6, vcc, %6, %7, vcc\n v_addc_co_u32 %7, vcc, %7, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(*(uint32_t*)&c4), "+v"(*(uint32_t*)&c5), "+v"(*(uint32_t*)&c6), "+v"(*(uint32_t*)&c7) :: "vcc");) }
    if (OP == 7) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %1\n v_fma_f64 %1, %1, %8, %2\n v_fma_f64 %2, %2, %8, %3\n v_fma_f64 %3, %3, %8, %0\n v_fma_f64 %4, %4, %8, %5\n v_fma_f64 %5, %5, %8, %6\n v_fma_f64 %6, %6, %8, %7\n v_fma_f64 %7, %7, %8, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1.0000001));) }
    if (OP == 8) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(*(uint32_t*)&c4), "+v"(*(uint32_t*)&c5), "+v"(*(uint32_t*)&c6), "+v"(*(uint32_t*)&c7));) }
    if (OP == 9) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %4\n v_add3_u32 %7, %7, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(*(uint32_t*)&c4), "+v"(*(uint32_t*)&c5), "+v"(*(uint32_t*)&c6), "+v"(*(uint32_t*)&c7));) }
    if (OP == 10) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1.0000001));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) + (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int OP> void run(const char* name, int waves_per_simd) {
  uint32_t* out; hipMalloc(&out, 4 << 20);
  int cus = 256; int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = 1 wave/SIMD per block
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)ITER * 64;   // 8 x 8 per iteration
  double ns_per_instr = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-18s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.1f cycles @2.4GHz)  chip %.2f Tinstr-lanes/s\n", name, waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4, 1024.0 * 64 / ns_per_instr / 1e3);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_mad_u64_u32", w); run<1>("v_mul_lo_u32", w); run<2>("v_mul_hi_u32", w); run<3>("v_mad_u32_u24", w);
    run<4>("v_mul_hi_u32_u24", w); run<5>("v_lshl_add_u64", w); run<6>("v_add(c)_co_u32", w); run<7>("v_fma_f64", w);
    run<10>("v_mul_f64", w); run<8>("v_mov_b32", w); run<9>("v_add3_u32", w);
  }
  return 0;
}
