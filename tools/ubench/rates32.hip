// VALU issue rates of the plain 32-bit (and a few 64-bit) bookkeeping instructions of the Montgomery rows, next to
// v_mad_u64_u32 — round 3: are simple 32-bit ops issued at 2 cycles per wave on gfx950's SIMD-32 (as v_mov_b32 / v_fma_f32)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define REP8(x) x x x x x x x x
// eight independent chains r0..r7, each instruction reads its own chain + a neighbour (no immediate folding possible)
#define CH3(op) { \
  REP8(asm volatile(op " %0, %0, %1\n " op " %1, %1, %2\n " op " %2, %2, %3\n " op " %3, %3, %0\n " op " %4, %4, %5\n " op " %5, %5, %6\n " op " %6, %6, %7\n " op " %7, %7, %4" \
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
#define CH4(op) { \
  REP8(asm volatile(op " %0, %0, %1, %2\n " op " %1, %1, %2, %3\n " op " %2, %2, %3, %0\n " op " %3, %3, %0, %1\n " op " %4, %4, %5, %6\n " op " %5, %5, %6, %7\n " op " %6, %6, %7, %4\n " op " %7, %7, %4, %5" \
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x9e3779b9, a4 = a0 + 9, a5 = a1 + 11, a6 = a2 + 13, a7 = a3 + 17;
  uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a4, c5 = a5, c6 = a6, c7 = a7;
  for (int i = 0; i < ITER; ++i) {
    if (OP == 0) CH3("v_add_u32")
    if (OP == 1) CH3("v_and_b32")
    if (OP == 2) CH3("v_lshrrev_b32")
    if (OP == 3) CH3("v_xor_b32")
    if (OP == 4) CH3("v_sub_u32")
    if (OP == 5) CH4("v_lshl_or_b32")
    if (OP == 6) CH4("v_and_or_b32")
    if (OP == 7) CH4("v_alignbit_b32")
    if (OP == 8) CH4("v_lshl_add_u32")
    if (OP == 9) CH4("v_add_lshl_u32")
    if (OP == 10) CH4("v_bfe_u32")
    if (OP == 11) CH4("v_perm_b32")
    if (OP == 12) CH4("v_add3_u32")
    if (OP == 13) CH4("v_mad_u32_u24")
    if (OP == 14) CH3("v_mul_lo_u32")
    if (OP == 15) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %1, vcc, %1, %2\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %3, vcc, %3, %0\n v_add_co_u32 %4, vcc, %4, %5\n v_add_co_u32 %5, vcc, %5, %6\n v_add_co_u32 %6, vcc, %6, %7\n v_add_co_u32 %7, vcc, %7, %4"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");) }
    if (OP == 16) { REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %0, vcc\n v_addc_co_u32 %4, vcc, %4, %5, vcc\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_addc_co_u32 %6, vcc, %6, %7, vcc\n v_addc_co_u32 %7, vcc, %7, %4, vcc"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");) }
    if (OP == 17) { REP8(asm volatile("v_lshrrev_b64 %0, 28, %1\n v_lshrrev_b64 %1, 28, %2\n v_lshrrev_b64 %2, 28, %3\n v_lshrrev_b64 %3, 28, %0\n v_lshrrev_b64 %4, 28, %5\n v_lshrrev_b64 %5, 28, %6\n v_lshrrev_b64 %6, 28, %7\n v_lshrrev_b64 %7, 28, %4"
                      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));) }
    if (OP == 18) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %4\n v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0"
                      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));) }
    if (OP == 19) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
                      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(a1) : "vcc");) }
    if (OP == 20) { REP8(asm volatile("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n v_mad_u64_u32 %1, s[10:11], %8, %9, %1\n v_mad_u64_u32 %2, s[10:11], %8, %9, %2\n v_mad_u64_u32 %3, s[10:11], %8, %9, %3\n v_mad_u64_u32 %4, s[10:11], %8, %9, %4\n v_mad_u64_u32 %5, s[10:11], %8, %9, %5\n v_mad_u64_u32 %6, s[10:11], %8, %9, %6\n v_mad_u64_u32 %7, s[10:11], %8, %9, %7"
                      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(a1) : "s10", "s11");) }
    if (OP == 21) CH3("v_pk_add_u16")
    if (OP == 22) CH4("v_pk_mad_u16")
    if (OP == 23) CH4("v_mad_u32_u16")
    if (OP == 24) CH4("v_dot4_u32_u8")
    if (OP == 25) CH3("v_mul_u32_u24")
    if (OP == 26) CH3("v_mul_hi_u32")
    if (OP == 27) CH4("v_fma_f32")
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7);
}
template <int OP> void run(const char* name, int waves_per_simd) {
  uint32_t* out; hipMalloc(&out, 4 << 20);
  int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns = ms * 1e6 / ((double)ITER * 64 * waves_per_simd);
  printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.1f cycles @2.4GHz)\n", name, waves_per_simd, ms, ns, ns * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {2, 4}) {
    run<0>("v_add_u32", w); run<1>("v_and_b32", w); run<2>("v_lshrrev_b32", w); run<3>("v_xor_b32", w); run<4>("v_sub_u32", w);
    run<5>("v_lshl_or_b32", w); run<6>("v_and_or_b32", w); run<7>("v_alignbit_b32", w); run<8>("v_lshl_add_u32", w); run<9>("v_add_lshl_u32", w);
    run<10>("v_bfe_u32", w); run<11>("v_perm_b32", w); run<12>("v_add3_u32", w); run<13>("v_mad_u32_u24", w); run<14>("v_mul_lo_u32", w);
    run<15>("v_add_co_u32", w); run<16>("v_addc_co_u32", w); run<17>("v_lshrrev_b64", w); run<18>("v_lshl_add_u64", w);
    run<19>("v_mad_u64_u32 vcc", w); run<20>("v_mad_u64_u32 sgpr", w); run<21>("v_pk_add_u16", w); run<22>("v_pk_mad_u16", w);
    run<23>("v_mad_u32_u16", w); run<24>("v_dot4_u32_u8", w); run<25>("v_mul_u32_u24", w); run<26>("v_mul_hi_u32", w);
    run<27>("v_fma_f32", w);
  }
  return 0;
}
