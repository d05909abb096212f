#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
  const size_t n = 16 * ((1u << 20) + 6);
  std::vector<uint32_t> k(n), v(n);
  std::mt19937 rng(1);
  for (size_t i = 0; i < n; ++i) { k[i] = rng() & 0x7fff; v[i] = (uint32_t)i; }
  uint32_t *dk, *dv, *ok, *ov; void* tmp = nullptr; size_t tb = 0;
  hipMalloc(&dk, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&ok, n * 4); hipMalloc(&ov, n * 4);
  hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
  rocprim::radix_sort_pairs(tmp, tb, dk, ok, dv, ov, n, 0, 15);
  hipMalloc(&tmp, tb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    rocprim::radix_sort_pairs(tmp, tb, dk, ok, dv, ov, n, 0, 15);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rocprim radix_sort_pairs n=%zu bits 0..15: %.3f ms (temp %zu B)\n", n, ms, tb);
  }
  // keys only, 64-bit packed
  return 0;
}
