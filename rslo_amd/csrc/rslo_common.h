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

// Tuning switches of the launch code (rslo_tuning_set / rslo_tuning_get, include/rslo_hip.h).  The library reads NO
// environment variable: tile shapes and variants that exist for A/B measurements and for the parity tests that pin
// every tiling against the oracle are set explicitly by the caller; the defaults (tuning.hip) are the measured choices.
enum RsloTune {
  RSLO_TUNE_CONV2D_WGRAD_S2_FULLRES,   // 1: full-resolution stride-2 weight gradient on the hand-written kernel
  RSLO_TUNE_CONV2D_WGRAD_NB,           // 16-channel blocks per wave of the dense weight gradient (2 | 4)
  RSLO_TUNE_CONV2D_WGRAD_WGS,          // workgroup target of its slab split
  RSLO_TUNE_CONV2D_FWD_TR,             // dense forward: rows per tile (0 = default 4; 4 | 8)
  RSLO_TUNE_CONV2D_FWD_MTW,            //   16-channel blocks per wave (0 = default 1; 1 | 2)
  RSLO_TUNE_CONV2D_FWD_OCC,            //   waves per SIMD the one-tap-ahead variants are compiled for (0 = default kernels)
  RSLO_TUNE_CONV2D_FWD_KC,             //   wave sets per workgroup (0 = choose; 1 | 2)
  RSLO_TUNE_CONV2D_FWD_LEAN,           //   96-register one-tap-ahead variant (-1 = choose; 0 | 1)
  RSLO_TUNE_CONV2D_FWD_XSC,            //   XCD channel classes (0 = choose, -1 = plain grid; 1 | 2 | 4 | 8)
  RSLO_TUNE_CONV2D_S2_MTW,             // stride-2 kernels: 16-channel blocks per wave (0 = choose; 1 | 2)
  RSLO_TUNE_CONV2D_S2_XSC,             //   XCD channel classes, as above
  RSLO_TUNE_BN_SMALL_RC,               // 1: register-cached single-launch BatchNorm on small maps; 0: two-pass loops
  RSLO_TUNE_SPCONV_RBW,                // k_spconv_v6: 16-row blocks per tile (0 = choose; 1 | 2)
  RSLO_TUNE_SPCONV_KS,                 //   waves per tile (0 = choose; 1 | 2 | 4)
  RSLO_TUNE_SPCONV_V,                  // fp32-MFMA sparse forward: 0 = choose, 100 + RBW forces v3's row blocking, 1..99 k_spconv
  RSLO_TUNE_SPCONV_WGRAD_SPLIT,        // 1: split-bf16 sparse weight gradient (k_wgrad3) on 32/64 channels; 0: fp32 MFMA
  RSLO_TUNE_WGRAD_XCD,                 // 1: XCD-ordered grid of the sparse weight gradients; 0: (chunk, offset) grid
  RSLO_TUNE_VFE_LDS,                   // 1: LDS-staged VFE mean; 0: one thread per voxel from memory
  RSLO_TUNE_CHAMFER,                   // 0: choose by size; 1: exhaustive; 2: pruned grid search
  RSLO_TUNE_CHAMFER_SEGMENTS,          // pruned search: target segments per query wave (0 = choose; 1..8)
  RSLO_TUNE_DENSE_TILED,               // 1 (default): dense() scatter / gather through 64 x 64 LDS tiles; 0: one thread per element
  RSLO_TUNE_CONV1X1_SPLIT,             // 1 (default): 1x1 output convolutions (cin <= 64) with a pixel's channels dealt to four waves; 0: one thread per pixel
  RSLO_TUNE_CONV2D_FWD_WL,             // dense 3x3 stride-1 forward / data gradient with the weight operands through LDS (k_conv2d_wl, measured slower): 0 (default) never; 1: 8-row tiles; 3: 6-row tiles
  RSLO_TUNE_COUNT
};
extern int g_rslo_tune[RSLO_TUNE_COUNT];
static inline int rslo_tune(RsloTune t) { return g_rslo_tune[t]; }

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
