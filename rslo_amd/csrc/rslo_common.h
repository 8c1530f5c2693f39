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

// ---------------------------------------------------------------------------------------
// Exact three-way bf16 split of fp32 values with ROUND-TO-NEAREST pieces (v_cvt_pk_bf16_f32 on gfx950):
//   x = h + m + l exactly,  |m| <= 2^-9 |x|,  |l| <= 2^-18 |x|
// (x - h and x - h - m are exact in fp32; the last remainder has <= 7 significant bits, so l is exact too).  The six
// bf16 products hh, hm, mh, hl, mm, lh then leave out only ml + lm + ll <= 2^-26 |x||y| with ZERO MEAN.  A split by
// truncation (mask the upper 16 bits) leaves remainders that all carry the sign of x: its dropped terms are 8x larger
// and add up coherently over a contraction -- measured through the whole network that cost a decimal digit of
// gradient accuracy against a float64 run (scripts/parity_report.py).  Cost is the same: one pack-convert replaces
// one v_and + v_perm per pair.
// ---------------------------------------------------------------------------------------
typedef __bf16 rslo_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rslo_f32x2 __attribute__((ext_vector_type(2)));

// packed bf16 pair (x0 in the low half, x1 in the high half), round to nearest even
__device__ __forceinline__ unsigned rslo_pk_bf16_rn(float x0, float x1) {
  const rslo_f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rslo_bf16x2));
}

struct RsloSplit2 {
  unsigned h, m, l;     // packed pairs: element 0 in the low half
};

__device__ __forceinline__ RsloSplit2 rslo_split2(float x0, float x1) {
  RsloSplit2 o;
  o.h = rslo_pk_bf16_rn(x0, x1);
  const float r0 = x0 - __uint_as_float(o.h << 16), r1 = x1 - __uint_as_float(o.h & 0xffff0000u);
  o.m = rslo_pk_bf16_rn(r0, r1);
  const float s0 = r0 - __uint_as_float(o.m << 16), s1 = r1 - __uint_as_float(o.m & 0xffff0000u);
  o.l = rslo_pk_bf16_rn(s0, s1);
  return o;
}

// one value -> the three pieces as bf16 bit patterns (weight pre-split kernels)
__device__ __forceinline__ void rslo_split1(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
  const RsloSplit2 s = rslo_split2(x, 0.f);
  h = (unsigned short)(s.h & 0xffffu);
  m = (unsigned short)(s.m & 0xffffu);
  l = (unsigned short)(s.l & 0xffffu);
}

static inline int rslo_log2_i64(int64_t v) {
  int l = 0;
  while ((int64_t(1) << l) < v) ++l;
  return l;
}
