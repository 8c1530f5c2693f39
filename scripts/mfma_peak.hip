// Matrix-core issue-rate probe for gfx950: N back-to-back v_mfma_f32_16x16x32_bf16 on NA independent accumulators per
// wave, W waves per SIMD.  Prints cycles per MFMA per SIMD and the implied chip TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NA>
__global__ __launch_bounds__(256) void k_probe(int iters, float *out, long long *cyc) {
  u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b = {5u, threadIdx.x, 7u, 8u};
  f32x4 acc[NA];
  for (int i = 0; i < NA; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NA>
void run(int blocks_per_cu, int iters) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int nblk = p.multiProcessorCount * blocks_per_cu;
  float *out; long long *cyc;
  hipMalloc(&out, sizeof(float) * nblk * 256);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe<NA>, dim3(nblk), dim3(256), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_probe<NA>, dim3(nblk), dim3(256), 0, 0, iters, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = (double)nblk * 4 * iters * NA;          // MFMAs in total (4 waves per block)
  const double tf = mf * 16384.0 / (ms * 1e-3) / 1e12;
  printf("NA=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s bf16, wave clock64 cycles per MFMA %.2f (clock64 ticks, not shader clocks)\n", NA,
         blocks_per_cu, ms, tf, (double)c / ((double)iters * NA));
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  run<1>(1, 20000); run<2>(1, 20000); run<4>(1, 20000); run<8>(1, 20000);
  run<4>(2, 20000); run<8>(2, 20000);
  return 0;
}
