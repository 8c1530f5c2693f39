// Shared helpers for the gfx950 kernels behind include/rslo_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rslo_hip.h"

#define RSLO_EMPTY_KEY 0xFFFFFFFFu

extern "C" void rslo_set_error(const char *fmt, ...);

#define RSLO_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      rslo_set_error(__VA_ARGS__);       \
      return RSLO_EINVAL;                \
    }                                    \
  } while (0)

#define RSLO_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      rslo_set_error("%s: %s", name, hipGetErrorString(e_));                 \
      return RSLO_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

#define RSLO_HIP(call)                                                       \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      rslo_set_error("%s: %s", #call, hipGetErrorString(e_));                \
      return RSLO_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

static inline int64_t rslo_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct Dims3 {
  int d, h, w;
};
struct Int3 {
  int a, b, c;
};

__device__ __forceinline__ uint32_t rslo_lin(int b, int z, int y, int x, Dims3 s) {
  return ((uint32_t)((b * s.d + z) * s.h + y)) * (uint32_t)s.w + (uint32_t)x;
}

// Fibonacci hash of the 32-bit linear index into a power-of-two table.
__device__ __forceinline__ uint32_t rslo_hslot(uint32_t key, int shift) {
  return (key * 0x9E3779B1u) >> shift;
}

__device__ __forceinline__ int32_t rslo_hfind(const uint32_t *__restrict__ keys,
                                              const int32_t *__restrict__ vals, uint32_t mask,
                                              int shift, uint32_t key) {
  uint32_t s = rslo_hslot(key, shift);
  while (true) {
    uint32_t k = keys[s];
    if (k == key) return vals[s];
    if (k == RSLO_EMPTY_KEY) return -1;
    s = (s + 1) & mask;
  }
}

static inline int rslo_log2_i64(int64_t v) {
  int l = 0;
  while ((int64_t(1) << l) < v) ++l;
  return l;
}
