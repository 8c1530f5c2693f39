#include <string.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
int main() {
  const size_t n = 10000; const int bits = 38;
  std::vector<unsigned long long> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (i % 3 == 0) ? ~0ull : ((((unsigned long long)(i * 7919 % 10400)) << 24) | i);
  unsigned long long *in, *out; hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
  hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
  size_t tmp = 0;
  rocprim::radix_sort_keys(nullptr, tmp, (unsigned long long *)nullptr, (unsigned long long *)nullptr, n, 0, bits, (hipStream_t)0);
  printf("tmp %zu\n", tmp);
  void *ws; hipMalloc(&ws, tmp + 4096);
  hipError_t e = rocprim::radix_sort_keys(ws, tmp, in, out, n, 0, (unsigned)bits, (hipStream_t)0);
  hipDeviceSynchronize();
  printf("err %d %s\n", (int)e, hipGetErrorString(hipGetLastError()));
  std::vector<unsigned long long> o(n);
  hipMemcpy(o.data(), out, n * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += o[i] != h[i];
  printf("mismatches %zu first out %llx expected %llx last out %llx\n", bad, o[0], h[0], o[n - 1]);
  return 0;
}
