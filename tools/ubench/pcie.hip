// Is the host link full duplex under the HIP runtime?  256 MiB pinned buffers: H2D alone, D2H alone, both at once on two
// streams, and both at once with a kernel running on a third stream (plonk_ntt_batch's pipeline, capi.hip).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void spin(uint32_t* p, int iters) {
  uint32_t v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
  p[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t B = 256u << 20;
  void *h1, *h2, *d1, *d2; uint32_t* dk;
  hipHostMalloc(&h1, B, hipHostMallocDefault); hipHostMalloc(&h2, B, hipHostMallocDefault);
  hipMalloc(&d1, B); hipMalloc(&d2, B); hipMalloc((void**)&dk, 4096);
  hipStream_t s1, s2, s3;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (int i = 0; i < 4; ++i) hipMemcpyAsync(d1, h1, B, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1);
    double t1 = now();
    for (int i = 0; i < 4; ++i) hipMemcpyAsync(h2, d2, B, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    double t2 = now();
    for (int i = 0; i < 4; ++i) { hipMemcpyAsync(d1, h1, B, hipMemcpyHostToDevice, s1); hipMemcpyAsync(h2, d2, B, hipMemcpyDeviceToHost, s2); }
    hipStreamSynchronize(s1); hipStreamSynchronize(s2);
    double t3 = now();
    for (int i = 0; i < 4; ++i) {
      hipMemcpyAsync(d1, h1, B, hipMemcpyHostToDevice, s1); hipMemcpyAsync(h2, d2, B, hipMemcpyDeviceToHost, s2);
      hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s3, dk, 20000);
    }
    hipStreamSynchronize(s1); hipStreamSynchronize(s2); hipStreamSynchronize(s3);
    double t4 = now();
    printf("rep %d: H2D %.1f GB/s | D2H %.1f GB/s | both at once: %.1f GB/s each way (%.2f ms per 256 MiB pair) | + kernel: %.2f ms per pair\n", rep,
           4 * B / (t1 - t0) / 1e9, 4 * B / (t2 - t1) / 1e9, 4 * B / (t3 - t2) / 1e9, (t3 - t2) / 4 * 1e3, (t4 - t3) / 4 * 1e3);
  }
  return 0;
}
