// Shared pieces of the dense convolution kernels of the BEV head (conv2d.hip, conv2d_planes.hip): operand types, the
// exact three-way operand split, the LDS-only workgroup barrier and the XCD-aware tile order.
#pragma once
#include "rslo_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(1))) unsigned char glb_u8;
typedef __attribute__((address_space(1))) u32x4 glb_u32x4;
typedef __attribute__((address_space(1))) float glb_f32;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
// a wave-uniform global address, forced into scalar registers
__device__ __forceinline__ const glb_u8 *c2f_uniform(const glb_u8 *p) {
  const uint64_t a = (uint64_t)p;
  return (const glb_u8 *)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                          (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
}
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

#define C2_THREADS 256
#define C2_WAVES 4
#define MFMA_BF16(A, B, C) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

struct Split3 {
  u32x4 h, m, l;
};

// 8 fp32 values -> three bf16x8 operands; bit e of `mask` keeps value e, cleared bits give zeros
__device__ __forceinline__ Split3 split_masked(const float (&v)[8], unsigned mask) {
  Split3 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const RsloSplit2 s = rslo_split2(((mask >> (2 * p)) & 1u) ? v[2 * p] : 0.f,
                                     ((mask >> (2 * p + 1)) & 1u) ? v[2 * p + 1] : 0.f);
    o.h[p] = s.h;
    o.m[p] = s.m;
    o.l[p] = s.l;
  }
  return o;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier and the fence
// drains vmcnt(0): every global load in flight (the next chunk's values, the next chunk's weight operands) would have
// to land before any wave passes -- the prefetches this kernel family relies on would be serialised at every barrier.
#define C2F_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Conv2dFwdGeom {
  int B, cin, cout, H, W;      // cin = contraction channels, cout = produced channels of THIS call
  int tiles_x, tiles_y;
  int xsc, npix, ny;           // XCD-aware workgroup order (conv2d_xcd_tile); xsc = 0: plain (pixel tile, channel group) grid
  const float *res;            // optional [B][cout][H][W] added to the result in the epilogue (out = conv + bias + res), or NULL
};

// Which (pixel tile bx, output-channel group by) a workgroup takes.  Workgroups of a one-dimensional grid are dealt to the
// 8 XCDs round-robin (id & 7) and each XCD has its own 4 MB L2, so the plain grid makes every XCD pull the WHOLE split
// weight operand (3.5 MB for 256 -> 256) and a scattered eighth of the pixel tiles through the fabric.  Here the 8 XCDs
// are arranged as xsc channel classes x (8 / xsc) pixel ranges: XCD x computes the channel groups by = x % xsc (mod xsc)
// of the contiguous pixel-tile range x / xsc, the channel groups of one pixel tile back to back (they stage the same
// halo).  Fabric traffic per launch ~ (8 / xsc) * weights + xsc * input; conv2d_xcd_split() picks xsc for the layer.
// The tiles themselves are unchanged: same bits.  Grid = 8 * (ny / xsc) * ceil(npix / (8 / xsc)) workgroups.
// Measured in the step (rocprof, same box): k_conv2d_fwd<4,1,true> 36.4 -> 32.7 us over its 58 launches, -0.2 ms per step.
// (The same idea on the weight-gradient kernels -- an XCD takes a run of pixel slabs with all their dW tiles -- changed
// nothing: 28.9 vs 29.4 us, not kept.)
__device__ __forceinline__ bool conv2d_xcd_tile_id(int id, int xsc, int npix, int ny, int &bx, int &by);
__device__ __forceinline__ bool conv2d_xcd_tile(int xsc, int npix, int ny, int &bx, int &by) {
  if (xsc == 0) {
    bx = blockIdx.x;
    by = blockIdx.y;
    return true;
  }
  return conv2d_xcd_tile_id((int)blockIdx.x, xsc, npix, ny, bx, by);
}
// the same mapping for an explicit workgroup slot `id` (id & 7 = the XCD it runs on; xsc >= 1)
__device__ __forceinline__ bool conv2d_xcd_tile_id(int id, int xsc, int npix, int ny, int &bx, int &by) {
  const int x = id & 7, j = id >> 3;
  const int sp = 8 / xsc, cg = x % xsc, pg = x / xsc, nyl = ny / xsc;
  const int p0 = (pg * npix) / sp, p1 = ((pg + 1) * npix) / sp;
  const int pl = j / nyl;
  bx = p0 + pl;
  by = cg + xsc * (j - pl * nyl);
  return bx < p1;
}
// channel-class count for a layer: the divisor of ny (<= 8, power of two) with the least fabric traffic; the environment
// variable `name` forces it for A/B runs (-1: plain grid)
static int conv2d_xcd_split(RsloTune which, int ny, double weight_bytes, double input_bytes) {
  const int env = rslo_tune(which);
  if (env < 0) return 0;
  int best = 1;
  double best_t = 0;
  for (int sc = 1; sc <= 8 && sc <= ny; sc *= 2) {
    if (ny % sc) break;
    const double t = (8 / sc) * weight_bytes + sc * input_bytes;
    if (sc == 1 || t < best_t) { best = sc; best_t = t; }
    if (env == sc) return sc;
  }
  return best;
}
static dim3 conv2d_xcd_grid(int xsc, int npix, int ny) {
  if (xsc == 0) return dim3((unsigned)npix, (unsigned)ny);
  const int sp = 8 / xsc;
  return dim3((unsigned)(8 * (ny / xsc) * ((npix + sp - 1) / sp)));
}
