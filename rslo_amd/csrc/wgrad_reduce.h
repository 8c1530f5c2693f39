// Second stage of the weight gradients (dense slab partials of conv2d.hip, sparse pair-chunk partials of spconv.hip): the
// per-layer reduce kernels and the ONE-launch form over many layers (wgrad_reduce.hip) share these block bodies, so a layer's
// gradient has the same bits whichever launch adds its partials -- fixed summation order, no atomics.
#pragma once
#include "rslo_common.h"

// ---- dense (k_conv2d_wgrad / _s1 partials, ws [slab][tile][ntap][16][co_t]) -------------------------------------------------
// block `bid` of n_main_blocks + ceil(cout / 32): 32 (tile, tap, ci, co) elements x 8 slab groups -- group sg adds slabs sg,
// sg + 8, ... in order, the 8 group sums are added in group order; writes dW in OIHW.  Blocks >= n_main_blocks: bias gradient.
__device__ __forceinline__ void wr_dense_block(int bid, const float *__restrict__ ws, int n_slabs, int n_tiles, int n_cout_tiles,
                                               int co_t, int cin, int cout, float *__restrict__ dW, int n_main_blocks,
                                               const float *__restrict__ bws, float *__restrict__ dbias, int ntap,
                                               float (*part)[32]) {
  const int per_tile = ntap * 16 * co_t;
  const int64_t n = (int64_t)n_tiles * per_tile;
  const int se = threadIdx.x & 31, sg = threadIdx.x >> 5;
  if (bid >= n_main_blocks) {          // bias gradient: 32 channels per block, same 8-group slab order
    const int c = (bid - n_main_blocks) * 32 + se;
    float t = 0.f;
    if (c < cout)
      for (int sl = sg; sl < n_slabs; sl += 8) t += bws[(int64_t)sl * cout + c];
    part[sg][se] = t;
    __syncthreads();
    if (sg == 0 && c < cout) {
      float t8 = part[0][se];
#pragma unroll
      for (int q = 1; q < 8; ++q) t8 += part[q][se];
      dbias[c] = t8;
    }
    return;
  }
  const int64_t e = (int64_t)bid * 32 + se;
  float s = 0.f;
  if (e < n) {
    int sl = sg;
    for (; sl + 24 < n_slabs; sl += 32) {
      const float a0 = ws[(int64_t)(sl + 0) * n + e], a1 = ws[(int64_t)(sl + 8) * n + e];
      const float a2 = ws[(int64_t)(sl + 16) * n + e], a3 = ws[(int64_t)(sl + 24) * n + e];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; sl < n_slabs; sl += 8) s += ws[(int64_t)sl * n + e];
  }
  part[sg][se] = s;
  __syncthreads();
  if (sg != 0 || e >= n) return;
  float t8 = part[0][se];
#pragma unroll
  for (int q = 1; q < 8; ++q) t8 += part[q][se];
  const int tile = (int)(e / per_tile), r = (int)(e - (int64_t)tile * per_tile);
  const int t = r / (16 * co_t), r2 = r - t * 16 * co_t;
  const int ci = r2 / co_t, co = r2 - ci * co_t;
  const int ct = tile / n_cout_tiles, ot = tile - ct * n_cout_tiles;
  dW[((int64_t)(ot * co_t + co) * cin + ct * 16 + ci) * ntap + t] = t8;
}

// ---- sparse (k_wgrad2 / k_wgrad3 partials, ws [chunk][K][cin cout]) ----------------------------------------------------------
// block (bx, k) of (ceil(cc / 32), K [+ 1]): row k < K adds the chunk partials of offset k in chunk order; the optional row K adds
// the bias-gradient partial rows bpart [n_bpart][cout] in row order (8 independent loads in flight, ordered adds).
__device__ __forceinline__ void wr_sparse_block(int bx, int k, const float *__restrict__ ws, const int32_t *__restrict__ koff,
                                                int K, int chunk, int cc, float *__restrict__ dW,
                                                const float *__restrict__ bpart, int n_bpart, int cout,
                                                float *__restrict__ dbias, float *red /* [256] */) {
  if (k == K) {
    if (bx != 0) return;
    const int cp = cout <= 16 ? 16 : (cout <= 32 ? 32 : 64);
    const int c = threadIdx.x % cp, part = threadIdx.x / cp, nparts = 256 / cp;
    float s = 0.f;
    if (c < cout) {
      for (int b = part; b < n_bpart; b += 8 * nparts) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int bb = b + u * nparts;
          v[u] = bb < n_bpart ? bpart[(int64_t)bb * cout + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (part == 0 && c < cout) {
      float t = 0.f;
      for (int q = 0; q < nparts; ++q) t += red[q * cp + c];
      dbias[c] = t;
    }
    return;
  }
  // 32 elements x 8 chunk groups per block: group sg adds chunks sg, sg + 8, ... in order, the 8 group sums are added
  // in group order (a serial loop over ~100 chunk partials per thread was a 16 us latency chain)
  float(*part)[32] = (float(*)[32])red;
  const int se = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int e = bx * 32 + se;
  const int n = koff[k + 1] - koff[k];
  const int nch = (n + chunk - 1) / chunk;
  float s = 0.f;
  if (e < cc) {
    int c = sg;
    for (; c + 24 < nch; c += 32) {
      const float a0 = ws[((int64_t)(c + 0) * K + k) * cc + e], a1 = ws[((int64_t)(c + 8) * K + k) * cc + e];
      const float a2 = ws[((int64_t)(c + 16) * K + k) * cc + e], a3 = ws[((int64_t)(c + 24) * K + k) * cc + e];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; c < nch; c += 8) s += ws[((int64_t)c * K + k) * cc + e];
  }
  part[sg][se] = s;
  __syncthreads();
  if (sg != 0 || e >= cc) return;
  float t = part[0][se];
#pragma unroll
  for (int q = 1; q < 8; ++q) t += part[q][se];
  dW[(int64_t)k * cc + e] = t;
}

// ---- deferral: the launch code hands a layer's reduce to the caller's sink instead of launching it (rslo_wgrad_reduce_defer) -----
extern thread_local RsloWgradReduce *g_wr_sink;
extern thread_local int g_wr_cap;
extern thread_local int *g_wr_count;
static inline bool wr_defer(const RsloWgradReduce &d) {
  if (!g_wr_sink || !g_wr_count || *g_wr_count >= g_wr_cap) return false;
  g_wr_sink[(*g_wr_count)++] = d;
  return true;
}
static inline RsloWgradReduce wr_dense_desc(const void *ws, int n_slabs, int n_tiles, int n_cout_tiles, int co_t, int cin,
                                            int cout, void *dW, int n_main, const void *bws, void *dbias, int ntap) {
  RsloWgradReduce d;
  d.kind = 0;
  d.n_blocks = n_main + (dbias ? (int)rslo_cdiv(cout, 32) : 0);
  d.ws = ws; d.dW = dW; d.aux = bws; d.dbias = dbias; d.koff = nullptr;
  d.p[0] = n_slabs; d.p[1] = n_tiles; d.p[2] = n_cout_tiles; d.p[3] = co_t; d.p[4] = cin; d.p[5] = cout; d.p[6] = ntap; d.p[7] = n_main;
  return d;
}
static inline RsloWgradReduce wr_sparse_desc(const void *ws, const void *koff, int K, int chunk, int cc, void *dW,
                                             const void *bpart, int n_bpart, int cout, void *dbias) {
  RsloWgradReduce d;
  const int gx = (int)rslo_cdiv(cc, 32);
  d.kind = 2;
  d.n_blocks = gx * (K + (dbias ? 1 : 0));
  d.ws = ws; d.dW = dW; d.aux = bpart; d.dbias = dbias; d.koff = koff;
  d.p[0] = K; d.p[1] = chunk; d.p[2] = cc; d.p[3] = n_bpart; d.p[4] = cout; d.p[5] = gx; d.p[6] = 0; d.p[7] = 0;
  return d;
}
