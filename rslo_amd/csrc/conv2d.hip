// Dense 2-D convolution weight gradient for the BEV head (3x3, stride 1 or 2, padding 1, NCHW fp32) on the bf16 matrix
// cores with exactly split operands -- the dense counterpart of k_wgrad3 (spconv.hip).
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} dout[b][co][y][x] * in[b][ci][S y + ky - 1][S x + kx - 1]
//
// is a GEMM whose contraction index runs over the OUTPUT pixels, which are contiguous in NCHW for both operands: a
// 32-pixel chunk of one image plane is one v_mfma_f32_16x16x32_bf16 step, lane (li = lane & 15, g = lane >> 4) holding
// pixels 8g..8g+7 of channel li -- two 16-byte loads, no transposition, no LDS staging.  The tap (ky, kx) only shifts
// the input window by (ky-1) W + (kx-1) floats (dword-aligned 16-byte loads) and masks the pixels whose neighbour falls
// outside the image.  Every fp32 operand value is split into three bf16 terms (hi + mid + lo, truncation splits, exact)
// and the six significant products are accumulated smallest first in fp32: fp32-accurate results at 6/16 of the fp32
// MFMA cycles.
//
// Work decomposition: a wave owns a [16 cin] x [16 NB cout] x 9-tap block of dW (36 accumulator tiles for NB = 4); the
// 4 waves of a workgroup take interleaved chunks of one pixel slab and are added in LDS in wave order; slab partials go
// to a workspace and are added in slab order by k_conv2d_wgrad_reduce (which also writes the OIHW layout): fixed
// summation order, no atomics, bit-reproducible run to run (MIOpen's split-K kernels for these shapes use atomics).
#include <stdlib.h>

#include "conv2d_tile.h"
#include "wgrad_reduce.h"


// 8 consecutive floats at base[idx .. idx+7]; lanes whose window leaves [0, total) read element-wise under the mask
__device__ __forceinline__ void load8(const float *__restrict__ base, int64_t idx, int64_t total, unsigned mask,
                                      float (&v)[8]) {
  if (idx >= 0 && idx + 8 <= total) {
    const f32x4u a = *(const f32x4u *)(base + idx);
    const f32x4u b = *(const f32x4u *)(base + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((mask >> e) & 1u) ? base[idx + e] : 0.f;
  }
}

// stride-2 input window: the 8 values in[idx + 2e]; 15 consecutive floats are loaded where they fit
__device__ __forceinline__ void load8_s2(const float *__restrict__ base, int64_t idx, int64_t total, unsigned mask,
                                         float (&v)[8]) {
  if (idx >= 0 && idx + 16 <= total) {
    const f32x4u a = *(const f32x4u *)(base + idx);
    const f32x4u b = *(const f32x4u *)(base + idx + 4);
    const f32x4u c = *(const f32x4u *)(base + idx + 8);
    const f32x4u d = *(const f32x4u *)(base + idx + 12);
    v[0] = a.x; v[1] = a.z; v[2] = b.x; v[3] = b.z;
    v[4] = c.x; v[5] = c.z; v[6] = d.x; v[7] = d.z;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((mask >> e) & 1u) ? base[idx + 2 * e] : 0.f;
  }
}

struct Conv2dGeom {
  int B, cin, cout, H, W;       // H, W: OUTPUT size (= input size for stride 1)
  int Hin, Win;                 // input size
  int cpi;                      // 32-pixel chunks per output image plane
  int chunks_per_slab;
  int n_cin_tiles, n_cout_tiles;
};

// grid (n_cin_tiles * n_cout_tiles, n_slabs); ws [slab][tile][NT][16][16 NB]
// NT = 9: the 3x3 / padding-1 layer; NT = 1: only the centre tap, which at stride 2 IS the 1x1 / padding-0 downsample layer
// (out[y][x] = W in[2y][2x]): its weight gradient from the same kernel
template <int NB, int STRIDE, int NT = 9>
__global__ __launch_bounds__(C2_THREADS, 2) void k_conv2d_wgrad(const float *__restrict__ in,
                                                             const float *__restrict__ dout, Conv2dGeom gm,
                                                             float *__restrict__ ws) {
  constexpr int CO_T = 16 * NB;
  __shared__ __attribute__((aligned(16))) float red[NT * 16 * CO_T];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int ct = tile / gm.n_cout_tiles, ot = tile - ct * gm.n_cout_tiles;
  const int ci0 = ct * 16, co0 = ot * CO_T;
  const int HW = gm.H * gm.W;
  const int64_t HWin = (int64_t)gm.Hin * gm.Win;
  const int64_t in_total = (int64_t)gm.B * gm.cin * HWin, out_total = (int64_t)gm.B * gm.cout * HW;
  const int n_chunks = gm.B * gm.cpi;
  const int c_begin = blockIdx.y * gm.chunks_per_slab;
  const int c_end = min(c_begin + gm.chunks_per_slab, n_chunks);

  f32x4 acc[NT][NB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c = c_begin + wid; c < c_end; c += C2_WAVES) {
    const int b = c / gm.cpi;
    const int p = (c - b * gm.cpi) * 32 + 8 * g;          // this lane's first output pixel in the plane
    int y = p / gm.W, x = p - (p / gm.W) * gm.W;
    const int y0 = y, x0 = x;
    // validity of the lane's 8 pixels per tap row / column
    unsigned vp = 0, vx0 = 0, vx2 = 0, vy0 = 0, vy2 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned bit = 1u << e;
      if (p + e < HW) vp |= bit;
      if (STRIDE * x - 1 >= 0) vx0 |= bit;
      if (STRIDE * x + 1 < gm.Win) vx2 |= bit;
      if (STRIDE * y - 1 >= 0) vy0 |= bit;
      if (STRIDE * y + 1 < gm.Hin) vy2 |= bit;
      if (++x == gm.W) { x = 0; ++y; }
    }
    // dout operands of all NB cout blocks
    u32x4 bh[NB], bm[NB], bl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float v[8];
      load8(dout, ((int64_t)b * gm.cout + co0 + 16 * nb + li) * HW + p, out_total, vp, v);
      const Split3 s = split_masked(v, vp);
      bh[nb] = s.h; bm[nb] = s.m; bl[nb] = s.l;
    }
    const int64_t plane = ((int64_t)b * gm.cin + ci0 + li) * HWin;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (NT == 1 && (ky != 1 || kx != 1)) continue;
        unsigned mask = vp;
        if (ky == 0) mask &= vy0;
        if (ky == 2) mask &= vy2;
        if (kx == 0) mask &= vx0;
        if (kx == 2) mask &= vx2;
        float v[8];
        if (STRIDE == 1) {
          load8(in, plane + p + (ky - 1) * gm.Win + (kx - 1), in_total, mask, v);
        } else {
          // the 8 output pixels may wrap to the next output row: input index is not affine across the wrap
          const int wrap = gm.W - x0;                       // first element that lies in the next output row (>= 8: none)
          if (wrap >= 8) {
            load8_s2(in, plane + (int64_t)(2 * y0 + ky - 1) * gm.Win + 2 * x0 + kx - 1, in_total, mask, v);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int yy = e < wrap ? y0 : y0 + 1, xx = e < wrap ? x0 + e : e - wrap;
              v[e] = ((mask >> e) & 1u) ? in[plane + (int64_t)(2 * yy + ky - 1) * gm.Win + 2 * xx + kx - 1] : 0.f;
            }
          }
        }
        const Split3 a = split_masked(v, mask);
        const int t = NT == 1 ? 0 : ky * 3 + kx;
        // six products per block, smallest first; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(a.l, bh[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(a.m, bm[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(a.h, bl[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(a.m, bh[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(a.h, bm[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(a.h, bh[nb], acc[t][nb]);
      }
    }
  }

  // waves are added in wave order through LDS: red[t][ci][co]
  for (int w = 0; w < C2_WAVES; ++w) {
    if (wid == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = (t * 16 + 4 * g + j) * CO_T + 16 * nb + li;
            float v = acc[t][nb][j];
            if (w) v += red[e];
            red[e] = v;
          }
    }
    __syncthreads();
  }
  float *dst = ws + ((int64_t)blockIdx.y * gridDim.x + tile) * (NT * 16 * CO_T);
  for (int e = tid; e < NT * 16 * CO_T; e += C2_THREADS) dst[e] = red[e];
}

// Stride-1 kernel: the three taps of one kernel row read the same 10-float window r[0..9] = in[p-1 .. p+8] of an input
// row, so the window is split ONCE (hi/mid/lo) and the three operands are assembled from two packings of it: even pairs
// E[i] = (r[2i], r[2i+1]) serve kx = 0 (E[0..3]) and kx = 2 (E[1..4]), odd pairs O[i] = (r[2i+1], r[2i+2]) serve
// kx = 1 -- 77 VALU operations per row instead of 180.  Border masks are formed from the row-wrap position (no
// per-pixel compares) and applied as bit masks on the packed operands, only in chunks where some lane needs them.
// The 4 waves are added through two LDS regions in two phases ((w0 + w2) + (w1 + w3), fixed order).
template <int NB, bool LP = false>
__global__ __launch_bounds__(C2_THREADS, NB == 2 ? 3 : 2) void k_conv2d_wgrad_s1(const float *__restrict__ in,
                                                                               const float *__restrict__ dout,
                                                                               Conv2dGeom gm, float invW,
                                                                               float *__restrict__ ws,
                                                                               float *__restrict__ bws) {
  constexpr int CO_T = 16 * NB, LDW = CO_T + 4;
  __shared__ __attribute__((aligned(16))) float red[2][9 * 16 * LDW];
  __shared__ float bred[C2_WAVES][CO_T];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int ct = tile / gm.n_cout_tiles, ot = tile - ct * gm.n_cout_tiles;
  const int ci0 = ct * 16, co0 = ot * CO_T;
  const int W = gm.W, H = gm.H, HW = H * W;
  const int64_t in_total = (int64_t)gm.B * gm.cin * HW, out_total = (int64_t)gm.B * gm.cout * HW;
  const int n_chunks = gm.B * gm.cpi;
  const int c_begin = blockIdx.y * gm.chunks_per_slab;
  const int c_end = min(c_begin + gm.chunks_per_slab, n_chunks);
  // bias gradient = per-channel sum of dout: the workgroups of cin tile 0 hold every dout value once as an operand
  const bool do_bias = bws != nullptr && ct == 0;
  float bsum[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bsum[nb] = 0.f;

  f32x4 acc[9][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c = c_begin + wid; c < c_end; c += C2_WAVES) {
    const int b = c / gm.cpi;
    const int p = (c - b * gm.cpi) * 32 + 8 * g;          // this lane's first pixel in the plane
    int y0 = (int)((float)p * invW);
    int x0 = p - y0 * W;
    if (x0 < 0) { x0 += W; --y0; }
    if (x0 >= W) { x0 -= W; ++y0; }
    const int wrap = W - x0;                              // element index of the first pixel of row y0 + 1 (>= 8: none)
    const unsigned first = wrap >= 8 ? 0xffu : ((1u << wrap) - 1u);                  // pixels in row y0
    const unsigned vp = p + 8 <= HW ? 0xffu : (p < HW ? ((1u << (HW - p)) - 1u) : 0u);
    const unsigned nx0 = ~((x0 == 0 ? 1u : 0u) | (wrap < 8 ? (1u << wrap) : 0u));    // x >= 1
    const unsigned nx2 = ~(wrap <= 8 ? (1u << (wrap - 1)) : 0u);                     // x <= W - 2
    const unsigned vy0 = y0 >= 1 ? 0xffu : (~first & 0xffu);                         // y >= 1
    const unsigned vy2 = (y0 <= H - 2 ? first : 0u) | (y0 <= H - 3 ? (~first & 0xffu) : 0u);   // y <= H - 2
    const bool slow = __any((int)((vp & nx0 & nx2 & vy0 & vy2 & 0xffu) != 0xffu));

    // dout operands of all NB cout blocks
    u32x4 bh[NB], bm[NB], bl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float v[8];
      load8(dout, ((int64_t)b * gm.cout + co0 + 16 * nb + li) * HW + p, out_total, vp, v);
      const Split3 s = split_masked(v, slow ? vp : 0xffu);
      bh[nb] = s.h; bm[nb] = s.m; bl[nb] = s.l;
      if (do_bias) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += ((vp >> e) & 1u) ? v[e] : 0.f;
        bsum[nb] += t;
      }
    }
    const int64_t plane = ((int64_t)b * gm.cin + ci0 + li) * HW;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int64_t idx = plane + p + (ky - 1) * W;        // r[j] = in[idx - 1 + j]
      float r[10];
      if (idx >= 1 && idx + 9 <= in_total) {
        const f32x4u a = *(const f32x4u *)(in + idx);
        const f32x4u bq = *(const f32x4u *)(in + idx + 4);
        r[0] = in[idx - 1];
        r[1] = a.x; r[2] = a.y; r[3] = a.z; r[4] = a.w;
        r[5] = bq.x; r[6] = bq.y; r[7] = bq.z; r[8] = bq.w;
        r[9] = in[idx + 8];
      } else {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const int64_t q = idx - 1 + j;
          r[j] = (q >= 0 && q < in_total) ? in[q] : 0.f;
        }
      }
      // even pairs (2i, 2i+1) by the packed round-to-nearest split; odd pairs (2i+1, 2i+2) = high half of even pair
      // i | low half of even pair i+1
      unsigned Eh[5], Em[5], El[5], Oh[4], Om[4], Ol[4];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const RsloSplit2 sp = rslo_split2(r[2 * i], r[2 * i + 1]);
        Eh[i] = sp.h;
        Em[i] = sp.m;
        El[i] = sp.l;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Oh[i] = __builtin_amdgcn_perm(Eh[i + 1], Eh[i], 0x05040302u);
        Om[i] = __builtin_amdgcn_perm(Em[i + 1], Em[i], 0x05040302u);
        Ol[i] = __builtin_amdgcn_perm(El[i + 1], El[i], 0x05040302u);
      }
      const unsigned my = vp & (ky == 0 ? vy0 : (ky == 2 ? vy2 : 0xffu));
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        u32x4 ah, am, al;
#pragma unroll
        for (int pd = 0; pd < 4; ++pd) {
          ah[pd] = kx == 0 ? Eh[pd] : (kx == 1 ? Oh[pd] : Eh[pd + 1]);
          am[pd] = kx == 0 ? Em[pd] : (kx == 1 ? Om[pd] : Em[pd + 1]);
          al[pd] = kx == 0 ? El[pd] : (kx == 1 ? Ol[pd] : El[pd + 1]);
        }
        if (slow) {
          const unsigned m = my & (kx == 0 ? nx0 : (kx == 2 ? nx2 : 0xffu));
#pragma unroll
          for (int pd = 0; pd < 4; ++pd) {
            const unsigned dm = ((0u - ((m >> (2 * pd)) & 1u)) & 0x0000ffffu) |
                                ((0u - ((m >> (2 * pd + 1)) & 1u)) & 0xffff0000u);
            ah[pd] &= dm; am[pd] &= dm; al[pd] &= dm;
          }
        }
        const int t = ky * 3 + kx;
        if constexpr (!LP) {      // LP (bf16 operands, C4): only hh -- the mid / lo pieces are dead code then
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(al, bh[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(am, bm[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(ah, bl[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(am, bh[nb], acc[t][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(ah, bm[nb], acc[t][nb]);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = MFMA_BF16(ah, bh[nb], acc[t][nb]);
      }
    }
  }

  // (w0 + w2) in region 0, (w1 + w3) in region 1, then region 0 + region 1 on the way out
  float *my_red = red[wid & 1];
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    if ((wid >> 1) == ph) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = (t * 16 + 4 * g + j) * LDW + 16 * nb + li;
            float v = acc[t][nb][j];
            if (ph) v = my_red[e] + v;
            my_red[e] = v;
          }
    }
    __syncthreads();
  }
  float *dst = ws + ((int64_t)blockIdx.y * gridDim.x + tile) * (9 * 16 * CO_T);
  for (int e = tid; e < 9 * 16 * CO_T; e += C2_THREADS) {
    const int row = e / CO_T, co = e - row * CO_T;
    dst[e] = red[0][row * LDW + co] + red[1][row * LDW + co];
  }
  if (do_bias) {                     // lane groups, then waves in wave order, one partial row per slab
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float t = bsum[nb];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (g == 0) bred[wid][16 * nb + li] = t;
    }
    __syncthreads();
    if (tid < CO_T)
      bws[(int64_t)blockIdx.y * gm.cout + co0 + tid] = ((bred[0][tid] + bred[1][tid]) + bred[2][tid]) + bred[3][tid];
  }
}

// the slab partials added in slab order (fixed summation order), dW in OIHW: block body in wgrad_reduce.h (shared with the
// one-launch form over many layers, rslo_wgrad_reduce_many)
__global__ __launch_bounds__(256) void k_conv2d_wgrad_reduce(const float *__restrict__ ws, int n_slabs, int n_tiles,
                                                             int n_cout_tiles, int co_t, int cin, int cout,
                                                             float *__restrict__ dW, int n_main_blocks,
                                                             const float *__restrict__ bws, float *__restrict__ dbias,
                                                             int ntap) {
  __shared__ float part[8][32];
  wr_dense_block((int)blockIdx.x, ws, n_slabs, n_tiles, n_cout_tiles, co_t, cin, cout, dW, n_main_blocks, bws, dbias, ntap, part);
}

static int conv2d_plan(int B, int cin, int cout, int H, int W, int stride, Conv2dGeom *gm, int *nb, int *n_slabs) {
  if (!(stride == 1 || stride == 2) || B <= 0 || H <= 0 || W < 8 || cin % 16 != 0 || cout % 32 != 0) return 0;
  // measured (scripts/bench_conv2d_wgrad.py): the stride-2 kernel re-reads a 15-float window per tap and loses to the
  // library on the full-resolution map (256->128 at 96x176: 170-210 us vs 131 us); smaller maps win (35-42 vs 49-57 us)
  // As leaf work on the weight-gradient stream (rslo_amd/streams.py) the difference no longer shows in the step, and the
  // hand-written kernel is bit-reproducible where the library's split-K kernels use atomics: it is the default;
  // conv2d_wgrad_s2_fullres = 0 hands the full-resolution layer back to the library.
  if (stride == 2 && (int64_t)H * W >= 96 * 176 && !rslo_tune(RSLO_TUNE_CONV2D_WGRAD_S2_FULLRES)) return 0;
  const int nb_pref = rslo_tune(RSLO_TUNE_CONV2D_WGRAD_NB);
  *nb = (cout % 64 == 0 && nb_pref == 4) ? 4 : 2;
  gm->B = B; gm->cin = cin; gm->cout = cout;
  gm->Hin = H; gm->Win = W;
  gm->H = stride == 1 ? H : (H - 1) / 2 + 1;          // (H + 2 - 3) / S + 1
  gm->W = stride == 1 ? W : (W - 1) / 2 + 1;
  if (gm->W < 8) return 0;
  gm->cpi = (int)rslo_cdiv((int64_t)gm->H * gm->W, 32);
  gm->n_cin_tiles = cin / 16;
  gm->n_cout_tiles = cout / (16 * *nb);
  const int tiles = gm->n_cin_tiles * gm->n_cout_tiles;
  const int n_chunks = B * gm->cpi;
  int target = rslo_tune(RSLO_TUNE_CONV2D_WGRAD_WGS);
  if (target < 1) target = 768;
  int s = (int)rslo_cdiv(target, tiles);
  const int max_s = n_chunks / 8 > 0 ? n_chunks / 8 : 1;   // at least 2 chunks per wave
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  gm->chunks_per_slab = (int)rslo_cdiv(n_chunks, s);
  *n_slabs = (int)rslo_cdiv(n_chunks, gm->chunks_per_slab);
  return 1;
}

extern "C" int rslo_conv2d_wgrad_supported(int cin, int cout, int H, int W, int stride) {
  Conv2dGeom gm;
  int nb, ns;
  return conv2d_plan(1, cin, cout, H, W, stride, &gm, &nb, &ns);
}

extern "C" size_t rslo_conv2d_wgrad_ws_bytes(int B, int cin, int cout, int H, int W, int stride) {
  Conv2dGeom gm;
  int nb, ns;
  if (!conv2d_plan(B, cin, cout, H, W, stride, &gm, &nb, &ns)) return 0;
  return (size_t)ns * gm.n_cin_tiles * gm.n_cout_tiles * 9 * 16 * 16 * nb * sizeof(float) +
         (size_t)ns * cout * sizeof(float);      // + one bias-gradient partial row per slab
}

static int conv2d_wgrad_launch(const float *in, const float *dout, int B, int cin, int cout, int H, int W, int stride,
                               float *dW, float *dbias, void *ws, size_t ws_bytes, void *stream, bool lp);

extern "C" int rslo_conv2d_wgrad(const float *in, const float *dout, int B, int cin, int cout, int H, int W,
                                 int stride, float *dW, float *dbias, void *ws, size_t ws_bytes, void *stream) {
  return conv2d_wgrad_launch(in, dout, B, cin, cout, H, W, stride, dW, dbias, ws, ws_bytes, stream, false);
}

extern "C" int rslo_conv2d_wgrad_bf16(const float *in, const float *dout, int B, int cin, int cout, int H, int W,
                                      int stride, float *dW, float *dbias, void *ws, size_t ws_bytes, void *stream) {
  return conv2d_wgrad_launch(in, dout, B, cin, cout, H, W, stride, dW, dbias, ws, ws_bytes, stream, stride == 1);
}

static int conv2d_wgrad_launch(const float *in, const float *dout, int B, int cin, int cout, int H, int W, int stride,
                               float *dW, float *dbias, void *ws, size_t ws_bytes, void *stream, bool lp) {
  Conv2dGeom gm;
  int nb, ns;
  RSLO_CHECK_ARG(conv2d_plan(B, cin, cout, H, W, stride, &gm, &nb, &ns),
                 "rslo_conv2d_wgrad: unsupported shape cin=%d cout=%d H=%d W=%d stride=%d", cin, cout, H, W, stride);
  RSLO_CHECK_ARG(ws_bytes >= rslo_conv2d_wgrad_ws_bytes(B, cin, cout, H, W, stride), "rslo_conv2d_wgrad: workspace too small");
  RSLO_CHECK_ARG((int64_t)B * cin * H * W < (int64_t(1) << 40), "rslo_conv2d_wgrad: tensor too large");
  RSLO_CHECK_ARG(!dbias || stride == 1, "rslo_conv2d_wgrad: the fused bias gradient needs stride 1");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = gm.n_cin_tiles * gm.n_cout_tiles;
  float *bws = dbias ? (float *)ws + (size_t)ns * tiles * 9 * 16 * 16 * nb : nullptr;
  const dim3 grid(tiles, ns);
#define C2_LAUNCH(NBv, Sv) \
  hipLaunchKernelGGL((k_conv2d_wgrad<NBv, Sv>), grid, dim3(C2_THREADS), 0, st, in, dout, gm, (float *)ws)
  if (stride == 1) {
    const float invW = 1.0f / (float)gm.W;
    if (lp && nb == 4)
      hipLaunchKernelGGL((k_conv2d_wgrad_s1<4, true>), grid, dim3(C2_THREADS), 0, st, in, dout, gm, invW, (float *)ws, bws);
    else if (lp)
      hipLaunchKernelGGL((k_conv2d_wgrad_s1<2, true>), grid, dim3(C2_THREADS), 0, st, in, dout, gm, invW, (float *)ws, bws);
    else if (nb == 4)
      hipLaunchKernelGGL((k_conv2d_wgrad_s1<4>), grid, dim3(C2_THREADS), 0, st, in, dout, gm, invW, (float *)ws, bws);
    else
      hipLaunchKernelGGL((k_conv2d_wgrad_s1<2>), grid, dim3(C2_THREADS), 0, st, in, dout, gm, invW, (float *)ws, bws);
  } else if (nb == 4) C2_LAUNCH(4, 2);
  else C2_LAUNCH(2, 2);
#undef C2_LAUNCH
  RSLO_CHECK_LAUNCH("k_conv2d_wgrad");
  const int64_t n = (int64_t)tiles * 9 * 16 * 16 * nb;
  const int n_main = (int)rslo_cdiv(n, 32);
  if (wr_defer(wr_dense_desc(ws, ns, tiles, gm.n_cout_tiles, 16 * nb, cin, cout, dW, n_main, bws, dbias, 9))) return RSLO_OK;
  hipLaunchKernelGGL(k_conv2d_wgrad_reduce, dim3((unsigned)(n_main + (dbias ? (int)rslo_cdiv(cout, 32) : 0))), dim3(256), 0,
                     st, (const float *)ws, ns, tiles, gm.n_cout_tiles, 16 * nb, cin, cout, dW, n_main,
                     (const float *)bws, dbias, 9);
  RSLO_CHECK_LAUNCH("k_conv2d_wgrad_reduce");
  return RSLO_OK;
}

// Weight gradient of the 1x1 / stride-2 / padding-0 downsample layers (dW [cout,cin,1,1]): the centre tap of the stride-2
// kernel.  in [B,cin,H,W], dout [B,cout,Ho,Wo], Ho = (H - 1) / 2 + 1.  Same slab partials + fixed-order reduction.
static int conv2d_plan_1x1s2(int B, int cin, int cout, int H, int W, Conv2dGeom *gm, int *nb, int *n_slabs) {
  if (B <= 0 || H <= 0 || W < 15 || cin % 16 != 0 || cout % 32 != 0) return 0;      // 8 or more output columns
  *nb = 2;
  gm->B = B; gm->cin = cin; gm->cout = cout; gm->Hin = H; gm->Win = W;
  gm->H = (H - 1) / 2 + 1; gm->W = (W - 1) / 2 + 1;
  gm->cpi = (int)rslo_cdiv((int64_t)gm->H * gm->W, 32);
  gm->n_cin_tiles = cin / 16; gm->n_cout_tiles = cout / 32;
  const int tiles = gm->n_cin_tiles * gm->n_cout_tiles, n_chunks = B * gm->cpi;
  int s = (int)rslo_cdiv(768, tiles);
  const int max_s = n_chunks / 8 > 0 ? n_chunks / 8 : 1;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  gm->chunks_per_slab = (int)rslo_cdiv(n_chunks, s);
  *n_slabs = (int)rslo_cdiv(n_chunks, gm->chunks_per_slab);
  return 1;
}

extern "C" int rslo_conv1x1s2_wgrad_supported(int cin, int cout, int H, int W) {
  Conv2dGeom gm;
  int nb, ns;
  return conv2d_plan_1x1s2(1, cin, cout, H, W, &gm, &nb, &ns);
}

extern "C" size_t rslo_conv1x1s2_wgrad_ws_bytes(int B, int cin, int cout, int H, int W) {
  Conv2dGeom gm;
  int nb, ns;
  if (!conv2d_plan_1x1s2(B, cin, cout, H, W, &gm, &nb, &ns)) return 0;
  return (size_t)ns * gm.n_cin_tiles * gm.n_cout_tiles * 16 * 16 * nb * sizeof(float);
}

extern "C" int rslo_conv1x1s2_wgrad(const float *in, const float *dout, int B, int cin, int cout, int H, int W, float *dW,
                                    void *ws, size_t ws_bytes, void *stream) {
  Conv2dGeom gm;
  int nb, ns;
  RSLO_CHECK_ARG(in && dout && dW && conv2d_plan_1x1s2(B, cin, cout, H, W, &gm, &nb, &ns),
                 "rslo_conv1x1s2_wgrad: unsupported shape cin=%d cout=%d H=%d W=%d", cin, cout, H, W);
  RSLO_CHECK_ARG(ws && ws_bytes >= rslo_conv1x1s2_wgrad_ws_bytes(B, cin, cout, H, W), "rslo_conv1x1s2_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = gm.n_cin_tiles * gm.n_cout_tiles;
  hipLaunchKernelGGL((k_conv2d_wgrad<2, 2, 1>), dim3(tiles, ns), dim3(C2_THREADS), 0, st, in, dout, gm, (float *)ws);
  RSLO_CHECK_LAUNCH("k_conv2d_wgrad(1x1)");
  const int64_t n = (int64_t)tiles * 16 * 16 * nb;
  const int n_main = (int)rslo_cdiv(n, 32);
  if (wr_defer(wr_dense_desc(ws, ns, tiles, gm.n_cout_tiles, 16 * nb, cin, cout, dW, n_main, nullptr, nullptr, 1))) return RSLO_OK;
  hipLaunchKernelGGL(k_conv2d_wgrad_reduce, dim3((unsigned)n_main), dim3(256), 0, st, (const float *)ws, ns, tiles,
                     gm.n_cout_tiles, 16 * nb, cin, cout, dW, n_main, (const float *)nullptr, (float *)nullptr, 1);
  RSLO_CHECK_LAUNCH("k_conv2d_wgrad_reduce(1x1)");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward / data gradient of the dense 3x3, stride-1, padding-1 convolution (NCHW fp32), same split-bf16 arithmetic.
//
//   out[b][m][y][x] = bias[m] + sum_{k, ky, kx} A[m][k][ky][kx] * in[b][k][y + ky - 1][x + kx - 1]
//
// forward: A = W (m = cout, k = cin); data gradient: A[m = cin][k = cout][ky][kx] = W[k][m][2 - ky][2 - kx], in = dout.
// The contraction index of the MFMA runs over 32 input channels, which are strided in NCHW, so a workgroup stages the
// halo tile ((TR + 2) x 18 pixels x 32 channels) ONCE per channel chunk through LDS: each thread loads the 8 channels
// of one pixel (lanes along x: coalesced), splits them into hi / mid / lo bf16 and writes three 16-byte pixel-major
// rows; all 9 taps then read their B operands (pixel = column, 8 channels per lane) with one ds_read_b128 per plane at
// a shifted pixel address -- every input value is split once per workgroup instead of once per tap.  A operands
// (weights) are split beforehand by k_conv2d_wsplit into the exact MFMA fragment order (16 bytes per lane and plane).
// Workgroup = 4 waves as 2 (output-channel halves) x 2 (row halves); out tile = 32 MTW channels x TR x 16 pixels.
// ---------------------------------------------------------------------------------------------------------------------
#define C2F_PXB 208      // bytes per staged pixel: 3 planes x 32 channels x 2 B + 16 B pad
// stride-1 kernels: 224 B.  ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): with
// lane = pixel + 16 * octet the 16-byte slots of a group are (S * pixel + octet) mod 16, S = stride / 16 -- S = 13 puts two
// lanes of every group on one slot (SQ_LDS_BANK_CONFLICT = 44 % of the LDS cycles of the 208-byte layout), S = 14 none
#ifndef C2F_PXB1
#define C2F_PXB1 224
#endif

// Ws [cin_k / 32][9][n_m / 16][3][64][8] bf16 from W [cout][cin][3][3]; transpose = 0: m = cout, k = cin (forward);
// transpose = 1: m = cin, k = cout, taps flipped (data gradient)
// element i of the operand block of a [cout][cin][T] weight, T = ntap taps (9: 3x3, 1: 1x1)
__device__ __forceinline__ void conv2d_wsplit_elem(const float *__restrict__ W, int cin, int cout, int ntap, int transpose,
                                                   unsigned short *__restrict__ Ws, int64_t i) {
  const int n_m = transpose ? cin : cout, n_k = transpose ? cout : cin;
  const int n_mt = n_m / 16;
  const int64_t n = (int64_t)n_k * ntap * n_m;
  if (i >= n) return;
  const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
  int64_t r = i >> 9;
  const int mt = (int)(r % n_mt); r /= n_mt;
  const int tap = (int)(r % ntap);
  const int chunk = (int)(r / ntap);
  const int m = mt * 16 + (lane & 15), k = chunk * 32 + 8 * (lane >> 4) + e;
  const float x = transpose ? W[((int64_t)k * cin + m) * ntap + (ntap - 1 - tap)] : W[((int64_t)m * cin + k) * ntap + tap];
  const int64_t base = ((((int64_t)chunk * ntap + tap) * n_mt + mt) * 3) * 512 + lane * 8 + e;
  rslo_split1(x, Ws[base], Ws[base + 512], Ws[base + 1024]);
}

__global__ void k_conv2d_wsplit(const float *__restrict__ W, int cin, int cout, int ntap, int transpose,
                                unsigned short *__restrict__ Ws) {
  conv2d_wsplit_elem(W, cin, cout, ntap, transpose, Ws, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// all layers of a model in one launch: grid (ceil(max_n / 256), 2 * n_layers); row 2 l + t splits layer l with
// transpose = t into desc[l].ws_fwd / ws_dgrad
__global__ void k_conv2d_wsplit_many(const RsloConv2dSplitDesc *__restrict__ desc) {
  const RsloConv2dSplitDesc d = desc[blockIdx.y >> 1];
  const int transpose = blockIdx.y & 1;
  conv2d_wsplit_elem(d.W, d.cin, d.cout, d.ntap == 1 ? 1 : 9, transpose,
                     (unsigned short *)(transpose ? d.ws_dgrad : d.ws_fwd), (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}


// KC = 2 (small maps): a second set of four waves takes every other channel chunk with its own staging buffer and the two
// accumulator sets meet through LDS (set 0 + set 1, a fixed order).  On the 12x22 / 24x44 maps a launch is less than one
// workgroup per CU and lasts as long as ONE workgroup's chain of n_chunks x (global load -> split -> LDS -> 9 taps),
// 2.7 us per chunk against 0.7 us of MFMAs: two half-length chains side by side on the CU halve it.
template <int TR, int MTW, bool FULLA, bool LP = false, int KC = 1, int OCC = 2 / KC>
__global__ __launch_bounds__(256 * KC, OCC) void k_conv2d_fwd(const float *__restrict__ in, const unsigned short *__restrict__ Ws,
                                                    const float *__restrict__ bias, Conv2dFwdGeom gm,
                                                    float *__restrict__ out) {
  constexpr int NTW = TR / 2, HR = TR + 2, NPX = HR * 18;
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[KC][NPX * C2F_PXB1];
  const int grp = KC == 1 ? 0 : (int)(threadIdx.x >> 8);       // wave set (wave-uniform)
  unsigned char *lds = lds_all[grp];
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wid & 1, wn = wid >> 1;
  int bx, by;
  if (!conv2d_xcd_tile(gm.xsc, gm.npix, gm.ny, bx, by)) return;
  const int tx = bx % gm.tiles_x; bx /= gm.tiles_x;
  const int ty = bx % gm.tiles_y;
  const int b = bx / gm.tiles_y;
  const int x0 = tx * 16, y0 = ty * TR;
  const int H = gm.H, W = gm.W;
  const int64_t HW = (int64_t)H * W;
  const int n_mt = gm.cout / 16;
  const int mt0 = by * 2 * MTW + wm * MTW;          // first 16-channel output block of this wave

  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging tasks of this thread: task = (channel octet o, pixel q), q fastest; NTASK rounds of 256 threads
  constexpr int NTASK = (NPX * 4 + 255) / 256;
  // A task's 8 channel values of chunk c sit at `in` + (c * 32 + j) * HW floats (uniform) + the task's own byte offset
  // tsrc[r] (one 32-bit register instead of a 64-bit pointer per task; -1 = outside the image)
  int tsrc[NTASK];
  int tdst[NTASK];
  const glb_u8 *const in_b = (const glb_u8 *)in;
#pragma unroll
  for (int r = 0; r < NTASK; ++r) {
    const int task = tid + r * 256;
    const int o = task / NPX, q = task - o * NPX;
    const int qy = q / 18, qx = q - qy * 18;
    const int y = y0 - 1 + qy, x = x0 - 1 + qx;
    const bool ok = task < NPX * 4 && y >= 0 && y < H && x >= 0 && x < W;
    tsrc[r] = ok ? (int)((((int64_t)b * gm.cin + 8 * o) * HW + (int64_t)y * W + x) * 4) : -1;
    tdst[r] = task < NPX * 4 ? q * C2F_PXB1 + o * 16 : -1;
  }
#define C2F_RAW(CH, J, OFF) (*(const glb_f32 *)(c2f_uniform(in_b + ((int64_t)(CH) * 32 + (J)) * HW * 4) + (unsigned)(OFF)))
  float raw[NTASK][8];
  const int n_chunks = gm.cin / 32;
#pragma unroll
  for (int r = 0; r < NTASK; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[r][j] = (tsrc[r] >= 0 && grp < n_chunks) ? C2F_RAW(grp, j, tsrc[r]) : 0.f;
  // weight operands are fetched one tap ahead (they come from L2 / Infinity Cache: the split weights of a whole model do
  // not stay in one XCD's L2 between layers); tap 0 of a chunk is requested before the staging barrier
  // FULLA (small maps, few MFMAs per tap): all 9 taps of a chunk are held in registers and each tap's registers are
  // refilled for the NEXT chunk right after its MFMAs -- a prefetch distance of 9 taps
  constexpr int NA = FULLA ? 9 : 1;
  u32x4 ah[NA][MTW], am[NA][MTW], al[NA][MTW], nh[MTW], nm[MTW], nl[MTW];
  const unsigned short *wbase = Ws + (int64_t)mt0 * 3 * 512 + lane * 8;
#define C2F_LOAD_A(CH, TAP, H_, M_, L_)                                                                  \
  _Pragma("unroll") for (int mt = 0; mt < MTW; ++mt) {                                                    \
    const unsigned short *wp = wbase + ((((int64_t)(CH) * 9 + (TAP)) * n_mt + mt) * 3) * 512;           \
    H_[mt] = *(const u32x4 *)(wp);                                                                        \
    if constexpr (!LP) {                                                                                  \
      M_[mt] = *(const u32x4 *)(wp + 512);                                                                \
      L_[mt] = *(const u32x4 *)(wp + 1024);                                                               \
    }                                                                                                     \
  }
  if (grp < n_chunks) {
    if (FULLA) {
#pragma unroll
      for (int t = 0; t < NA; ++t) { C2F_LOAD_A(grp, t, ah[t], am[t], al[t]) }
    } else {
      C2F_LOAD_A(grp, 0, ah[0], am[0], al[0])
    }
  }
  for (int chunk = grp; chunk < n_chunks + grp; chunk += KC) {      // both wave sets pass the same number of barriers
    const bool live = chunk < n_chunks;
    if (live) {
    // split the prefetched values of this chunk into LDS, then prefetch the next chunk's (in flight during the MFMAs)
#pragma unroll
    for (int r = 0; r < NTASK; ++r) {
      if (tdst[r] >= 0) {
        const Split3 s = split_masked(raw[r], 0xffu);
        unsigned char *dst = lds + tdst[r];
        *(u32x4 *)(dst) = s.h;
        if constexpr (!LP) {      // LP (bf16 operands, C4): only the round-to-nearest bf16 value of each activation
          *(u32x4 *)(dst + 64) = s.m;
          *(u32x4 *)(dst + 128) = s.l;
        }
      }
    }
    if (chunk + KC < n_chunks) {
#pragma unroll
      for (int r = 0; r < NTASK; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[r][j] = tsrc[r] >= 0 ? C2F_RAW(chunk + KC, j, tsrc[r]) : 0.f;
    }
    }
    C2F_LDS_BARRIER();      // LDS only: the prefetched global loads stay in flight across it
    if (live) {
    u32x4 pbh[NTW], pbm[NTW], pbl[NTW];
#define C2F_LOAD_B(KY, KX)                                                                               \
  _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) {                                                    \
    const lds_u8 *bp = bbase + ((nt + (KY)) * 18 + (KX)) * C2F_PXB1;     /* one address register + immediates */ \
    pbh[nt] = *(const lds_u32x4 *)(bp);                                                                   \
    if constexpr (!LP) {                                                                                  \
      pbm[nt] = *(const lds_u32x4 *)(bp + 64);                                                            \
      pbl[nt] = *(const lds_u32x4 *)(bp + 128);                                                           \
    }                                                                                                     \
  }
    const lds_u8 *bbase = (const lds_u8 *)lds + (wn * NTW * 18 + li) * C2F_PXB1 + g * 16;
    constexpr bool BPF = OCC < 5;      // >= 5 waves per SIMD: no second operand set, the other waves cover the LDS latency
    if constexpr (BPF) { C2F_LOAD_B(0, 0) }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tap = ky * 3 + kx;
        constexpr int ta = 0;
        const int ia = FULLA ? tap : ta;
        if (!FULLA) {
          if (tap < 8) {
            C2F_LOAD_A(chunk, tap + 1, nh, nm, nl)
          } else if (chunk + KC < n_chunks) {
            C2F_LOAD_A(chunk + KC, 0, nh, nm, nl)
          }
        }
        // pixel operands one tap ahead: the LDS reads of tap + 1 are in flight during this tap's MFMAs
        u32x4 bh[NTW], bm[NTW], bl[NTW];
        if constexpr (!BPF) { C2F_LOAD_B(ky, kx) }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          bh[nt] = pbh[nt];
          if constexpr (!LP) { bm[nt] = pbm[nt]; bl[nt] = pbl[nt]; }
        }
        if (BPF && tap < 8) {
          const int ky2 = (tap + 1) / 3, kx2 = (tap + 1) % 3;
          C2F_LOAD_B(ky2, kx2)
        }
        // six products per block, smallest first; consecutive MFMAs hit different accumulators (LP: the hh product only)
        if constexpr (!LP) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(al[ia][mt], bh[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(am[ia][mt], bm[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(ah[ia][mt], bl[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(am[ia][mt], bh[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(ah[ia][mt], bm[nt], acc[mt][nt]);
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(ah[ia][mt], bh[nt], acc[mt][nt]);
        if (FULLA) {
          if (chunk + KC < n_chunks) { C2F_LOAD_A(chunk + KC, tap, ah[ia], am[ia], al[ia]) }
        } else {
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            ah[0][mt] = nh[mt];
            if constexpr (!LP) { am[0][mt] = nm[mt]; al[0][mt] = nl[mt]; }
          }
        }
      }
    }
    }
    C2F_LDS_BARRIER();      // LDS only: the prefetched global loads stay in flight across it
  }
#undef C2F_LOAD_A
#undef C2F_LOAD_B
#undef C2F_RAW
  if constexpr (KC == 2) {      // set 1 -> LDS (its own staging buffer, free after the last barrier) -> set 0
    float *slot = reinterpret_cast<float *>(lds_all[1]);
    if (grp == 1) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) slot[((mt * NTW + nt) * 4 + j) * 256 + tid] = acc[mt][nt][j];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt][nt][j] += slot[((mt * NTW + nt) * 4 + j) * 256 + tid];
  }

  const int x = x0 + li;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = (mt0 + mt) * 16 + 4 * g + j;
      const float bv = bias ? bias[m] : 0.f;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int y = y0 + wn * NTW + nt;
        if (x < W && y < H) {
          const int64_t o = ((int64_t)b * gm.cout + m) * HW + (int64_t)y * W + x;
          float v = acc[mt][nt][j] + bv;
          if (gm.res) v += gm.res[o];      // the residual branch's gradient of a BasicBlock: same bits as a separate add
          out[o] = v;
        }
      }
    }
}


#define g_c2f_occ rslo_tune(RSLO_TUNE_CONV2D_FWD_OCC)      // waves per SIMD of the one-tap-ahead variants (0: default kernels)
static int conv2d_fwd_plan(int B, int cin, int cout, int H, int W, int *tr, int *mtw) {
  if (B <= 0 || H <= 0 || W <= 0 || cin % 32 != 0 || cout % 32 != 0) return 0;
  if ((int64_t)B * cin * H * W * 4 >= ((int64_t)1 << 31)) return 0;      // per-lane input offsets are 31-bit byte offsets
  const int cfg_tr = rslo_tune(RSLO_TUNE_CONV2D_FWD_TR), cfg_mtw = rslo_tune(RSLO_TUNE_CONV2D_FWD_MTW);
  // measured inside the training step (profiles/README.md): 4-row tiles, one 16-channel block per wave and the whole
  // chunk's weight operands prefetched 9 taps ahead win on every map size of the head (45 vs 67 us on 48x88, 23 vs 54 us
  // on 12x22 against the 8-row / one-tap-ahead configurations, which stay selectable for experiments)
  int t = 4, m = 1;
  if (cfg_tr == 4 || cfg_tr == 6 || cfg_tr == 8) t = cfg_tr;
  if ((cfg_mtw == 1 || cfg_mtw == 2) && cout % (32 * cfg_mtw) == 0) m = cfg_mtw;
  // few input channels feeding many output channels on a full-resolution map (the data gradient of the decoder's
  // 192 -> 64 layer: 64 -> 192 at 96x176): two channel blocks per wave halve the staging per product; measured round 4
  // (scripts/sweep_conv2d_fwd.py, B = 4): 92.2 -> 83.3 us.  Every other head shape is slower that way.
  else if (cfg_mtw == 0 && cfg_tr == 0 && cin <= 64 && cout >= 192 && cout % 64 == 0 && (int64_t)H * W >= 96 * 176) m = 2;
  // 6-row tiles (conv2d_fwd_tr = 6, round 5): a wave covers 16 channels x 48 pixels -- 2/3 of the weight-operand bytes per
  // MFMA -- and 128 -> 128 at 48x88 becomes 768 workgroups = exactly 3 per CU.  Back-to-back launches win 5-12 % on the
  // 48x88 / 96x176 maps (scripts/sweep_conv2d_fwd.py: 36.9 -> 33.1, 66.1 -> 57.9, 34.2 -> 32.6, 91.1 -> 83.4 us), the training
  // step does not (10.84-10.99 ms either way in six alternating runs: at 3 waves per SIMD the kernel is the one that yields
  // to the leaf / covariance streams' kernels), so the default stays 4 rows.  Same bits.
  *tr = t; *mtw = m;
  return 1;
}

extern "C" int rslo_conv2d_fwd_supported(int cin, int cout, int H, int W) {
  int t, m;
  return conv2d_fwd_plan(1, cin, cout, H, W, &t, &m);
}

extern "C" size_t rslo_conv2d_wsplit_bytes(int cin, int cout) { return (size_t)3 * 9 * cin * cout * sizeof(unsigned short); }

extern "C" int rslo_conv2d_wsplit_k(const float *W, int cin, int cout, int ksize, int transpose, void *Ws, void *stream) {
  RSLO_CHECK_ARG(cin % 32 == 0 && cout % 32 == 0, "rslo_conv2d_wsplit: channels must be multiples of 32 (%d, %d)", cin, cout);
  RSLO_CHECK_ARG(ksize == 1 || ksize == 3, "rslo_conv2d_wsplit: kernel size must be 1 or 3");
  const int64_t n = (int64_t)cin * cout * ksize * ksize;
  hipLaunchKernelGGL(k_conv2d_wsplit, dim3((unsigned)rslo_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, W, cin, cout,
                     ksize * ksize, transpose, (unsigned short *)Ws);
  RSLO_CHECK_LAUNCH("k_conv2d_wsplit");
  return RSLO_OK;
}

extern "C" int rslo_conv2d_wsplit(const float *W, int cin, int cout, int transpose, void *Ws, void *stream) {
  return rslo_conv2d_wsplit_k(W, cin, cout, 3, transpose, Ws, stream);
}

extern "C" int rslo_conv2d_wsplit_many(const RsloConv2dSplitDesc *desc_dev, int n_layers, int64_t max_weight_elems,
                                       void *stream) {
  RSLO_CHECK_ARG(n_layers > 0 && n_layers < 32768 && max_weight_elems > 0, "rslo_conv2d_wsplit_many: bad sizes");
  hipLaunchKernelGGL(k_conv2d_wsplit_many, dim3((unsigned)rslo_cdiv(max_weight_elems, 256), (unsigned)(2 * n_layers)),
                     dim3(256), 0, (hipStream_t)stream, desc_dev);
  RSLO_CHECK_LAUNCH("k_conv2d_wsplit_many");
  return RSLO_OK;
}

static int conv2d_fwd_launch(const float *in, const void *Ws, const float *bias, int B, int cin, int cout, int H, int W,
                             float *out, void *stream, bool lp, const float *res = nullptr);

extern "C" int rslo_conv2d_fwd(const float *in, const void *Ws, const float *bias, int B, int cin, int cout, int H, int W,
                               float *out, void *stream) {
  return conv2d_fwd_launch(in, Ws, bias, B, cin, cout, H, W, out, stream, false);
}

extern "C" int rslo_conv2d_fwd_bf16(const float *in, const void *Ws, const float *bias, int B, int cin, int cout, int H,
                                    int W, float *out, void *stream) {
  return conv2d_fwd_launch(in, Ws, bias, B, cin, cout, H, W, out, stream, true);
}

extern "C" int rslo_conv2d_fwd_add(const float *in, const void *Ws, const float *bias, const float *res, int B, int cin,
                                   int cout, int H, int W, float *out, void *stream) {
  return conv2d_fwd_launch(in, Ws, bias, B, cin, cout, H, W, out, stream, false, res);
}

extern "C" int rslo_conv2d_fwd_add_bf16(const float *in, const void *Ws, const float *bias, const float *res, int B,
                                        int cin, int cout, int H, int W, float *out, void *stream) {
  return conv2d_fwd_launch(in, Ws, bias, B, cin, cout, H, W, out, stream, true, res);
}

// conv2d_wl.hip: the same convolution with the weight operands shared through LDS (bit-identical results)
int conv2d_wl_wanted(int B, int cin, int cout, int H, int W);
int conv2d_wl_launch(const float *in, const void *Ws, const float *bias, const float *res, int B, int cin, int cout, int H,
                     int W, float *out, void *stream);

static int conv2d_fwd_launch(const float *in, const void *Ws, const float *bias, int B, int cin, int cout, int H, int W,
                             float *out, void *stream, bool lp, const float *res) {
  int tr, mtw;
  RSLO_CHECK_ARG(conv2d_fwd_plan(B, cin, cout, H, W, &tr, &mtw), "rslo_conv2d_fwd: unsupported shape cin=%d cout=%d H=%d W=%d",
                 cin, cout, H, W);
  // explicit tile settings (experiments, the tiling tests) keep k_conv2d_fwd
  if (!lp && !rslo_tune(RSLO_TUNE_CONV2D_FWD_TR) && !rslo_tune(RSLO_TUNE_CONV2D_FWD_MTW) && !g_c2f_occ &&
      !rslo_tune(RSLO_TUNE_CONV2D_FWD_KC) && rslo_tune(RSLO_TUNE_CONV2D_FWD_LEAN) < 0 && conv2d_wl_wanted(B, cin, cout, H, W))
    return conv2d_wl_launch(in, Ws, bias, res, B, cin, cout, H, W, out, stream);
  Conv2dFwdGeom gm;
  gm.B = B; gm.cin = cin; gm.cout = cout; gm.H = H; gm.W = W;
  gm.res = res;
  gm.tiles_x = (int)rslo_cdiv(W, 16);
  gm.tiles_y = (int)rslo_cdiv(H, tr);
  // input bytes count the halo re-reads of the row tiles ((tr + 2) / tr); bf16-operand weights are one plane of three
  const double w_bytes = (lp ? 1.0 : 3.0) * 18.0 * cin * cout, in_bytes = 4.0 * B * cin * H * W * 1.5;
  gm.npix = B * gm.tiles_x * gm.tiles_y;
  gm.ny = cout / (32 * mtw);
  gm.xsc = conv2d_xcd_split(RSLO_TUNE_CONV2D_FWD_XSC, gm.ny, w_bytes, in_bytes);
  const dim3 grid = conv2d_xcd_grid(gm.xsc, gm.npix, gm.ny);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *ws = (const unsigned short *)Ws;
  // two wave sets per workgroup (channel chunks alternate between them) when the launch leaves CUs empty and the chain
  // is long (conv2d_fwd_kc = 1 | 2 forces it).  Measured (scripts/bench_conv2d_fwd.py): 256 -> 256 at 12x22 (192
  // workgroups, 8 chunks) 23.7 -> 19.0 us; 128 -> 128 at 24x44 (288 workgroups, 4 chunks) 17.1 -> 21.4 us, 512 -> 128 at
  // 24x44 50.9 -> 61.0 us, 48x88 and larger 44 -> 62 us: every wave streams its own weight operands (27 KB per chunk)
  // through the CU's vector L1, and a second wave set doubles that traffic wherever the CUs are already occupied
  const int kc_env = rslo_tune(RSLO_TUNE_CONV2D_FWD_KC);
  const int64_t wgs4 = (int64_t)B * gm.tiles_x * rslo_cdiv(H, 4) * (cout / 32);
  const bool kc2 = cin >= 64 && (kc_env ? kc_env == 2 : (cin >= 256 && wgs4 <= 200));
  if (lp) {       // bf16 operands (C4): the default tile configuration only
    gm.tiles_y = (int)rslo_cdiv(H, 4);
    gm.npix = B * gm.tiles_x * gm.tiles_y;
    gm.ny = cout / 32;
    gm.xsc = conv2d_xcd_split(RSLO_TUNE_CONV2D_FWD_XSC, gm.ny, w_bytes, in_bytes);
    const dim3 grid1 = conv2d_xcd_grid(gm.xsc, gm.npix, gm.ny);
    if (kc2) hipLaunchKernelGGL((k_conv2d_fwd<4, 1, true, true, 2>), grid1, dim3(512), 0, st, in, ws, bias, gm, out);
    else hipLaunchKernelGGL((k_conv2d_fwd<4, 1, true, true>), grid1, dim3(256), 0, st, in, ws, bias, gm, out);
    RSLO_CHECK_LAUNCH("k_conv2d_fwd(bf16)");
    return RSLO_OK;
  }
  // maps of >= 512 workgroups (2 or more per CU): the one-tap-ahead variant at 96 registers keeps 5 workgroups resident
  // per CU instead of 2 (20 waves hide the staging chain; the 9-tap weight prefetch is not needed with them; the 32-bit
  // input offsets above are what lets it fit with one spilled register).  Measured (scripts/conv2d_cfgs.sh, B = 4, the
  // 9-tap kernel -> 4 workgroups per CU at 104 registers -> 5): 128 -> 128 at 48x88 41.6 -> 38.0 -> 35.7 us, 64 -> 64 at
  // 96x176 44.1 -> 36.4 -> 35.2, 64 -> 192 at 96x176 114.8 -> 92.2 -> 87.8, 192 -> 64 103.8 -> 94.1 -> 89.7, 256 -> 64 at
  // 48x88 (576 workgroups) 47.0 -> 42.1; 128 -> 128 at 24x44 (288 workgroups) 16.5 -> 17.4 and 256 -> 256 at 12x22
  // 18.6 -> 25.0 keep the 9-tap kernel.  conv2d_fwd_lean = 0 / 1 forces it off / on.
  const int lean_env = rslo_tune(RSLO_TUNE_CONV2D_FWD_LEAN);
  if (tr == 4 && mtw == 1 && !kc2 && !g_c2f_occ && (lean_env < 0 ? wgs4 >= 512 : lean_env == 1)) {
    hipLaunchKernelGGL((k_conv2d_fwd<4, 1, false, false, 1, 5>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
    RSLO_CHECK_LAUNCH("k_conv2d_fwd(lean)");
    return RSLO_OK;
  }
  if (tr == 4 && g_c2f_occ) {
#define C2F_OCC(M, O) hipLaunchKernelGGL((k_conv2d_fwd<4, M, false, false, 1, O>), grid, dim3(256), 0, st, in, ws, bias, gm, out)
    if (mtw == 1 && g_c2f_occ == 3) C2F_OCC(1, 3);
    else if (mtw == 1 && g_c2f_occ == 4) C2F_OCC(1, 4);
    else if (mtw == 1 && g_c2f_occ == 4) C2F_OCC(1, 4);
    else if (mtw == 1) C2F_OCC(1, 5);
    else if (g_c2f_occ == 4) C2F_OCC(2, 4);
    else C2F_OCC(2, 3);
#undef C2F_OCC
  } else
  if (tr == 6 && g_c2f_occ == 4) hipLaunchKernelGGL((k_conv2d_fwd<6, 1, false, false, 1, 4>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  else if (tr == 6 && g_c2f_occ == 2) hipLaunchKernelGGL((k_conv2d_fwd<6, 1, true, false, 1, 2>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  else if (tr == 6) hipLaunchKernelGGL((k_conv2d_fwd<6, 1, false, false, 1, 3>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  else if (tr == 8 && mtw == 2) hipLaunchKernelGGL((k_conv2d_fwd<8, 2, false>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  else if (tr == 8) hipLaunchKernelGGL((k_conv2d_fwd<8, 1, false>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  else if (mtw == 2) hipLaunchKernelGGL((k_conv2d_fwd<4, 2, false>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  else if (kc2) hipLaunchKernelGGL((k_conv2d_fwd<4, 1, true, false, 2>), grid, dim3(512), 0, st, in, ws, bias, gm, out);
  else hipLaunchKernelGGL((k_conv2d_fwd<4, 1, true>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  RSLO_CHECK_LAUNCH("k_conv2d_fwd");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 layers of the BEV encoder (the first 3x3 convolution and the 1x1 downsample of every stage:
// rslo/models/odom_pred.py:398-426 -> custom_resnet_spc.BasicBlock / conv1x1): forward and data gradient on the same
// staged-halo, split-bf16 structure as k_conv2d_fwd.
//
//   forward        out[b][m][y][x]  = sum_{k, taps} A[m][k][tap] in[b][k][2y + ky - P][2x + kx - P]      (P = 1 / 0)
//   data gradient  din[b][m][Y][X]  = sum_{k, (ky,kx): Y = 2y + ky - P, X = 2x + kx - P} W[k][m][ky][kx] dout[b][k][y][x]
//
// The data gradient is computed per output PARITY CLASS (Y mod 2, X mod 2): a class's pixels (2r + py, 2c + px) form a
// dense grid over which the sum is a small stride-1 convolution of dout with 1, 2 or 4 of the 9 taps (3x3) or with the
// single tap / nothing (1x1: three of the four classes are zeros, written by the same launch).  No zero-stuffed
// intermediate, no atomics, no layout transposes (the library path: Winograd "dilation2" kernels + NCHW<->CNHW copies).
// One kernel: S_IN = sampling stride of the staged plane per class pixel (2: 3x3 forward; 1: data gradients and the 1x1
// forward, which stages only the sampled pixels); a class descriptor gives the taps (LDS offsets + index into the split
// weight operand) and where the class's pixels land in the output plane.  A workgroup stages its halo ONCE per channel
// chunk and computes all NCLS classes of its tile from it (data gradient: 4 classes, 9 tap products in total).
// ---------------------------------------------------------------------------------------------------------------------
struct Conv2dStrClass {
  int py, px;             // output pixel of class pixel (r, c): (s_out r + py, s_out c + px)
  int ny, nx;             // taps used (0: the class is identically zero)
  int oy[3], ox[3];       // halo offsets of the taps
  int wt[9];              // [iy * 3 + ix] -> tap index in the weight operand
  int rows, cols;         // class pixels per image
};
struct Conv2dStrGeom {
  int B, cin, cout;       // contraction / produced channels of THIS call
  int Hi, Wi, Ho, Wo;     // source plane, written plane
  int src_stride;         // staged pixel (qy, qx) = source pixel src_stride * (S_IN r0 + by + qy, ..) (2: 1x1 forward)
  int by, bx;             // halo origin offset
  int s_out, ntap_w;      // output stride of class pixels; taps per chunk in the weight operand (9 or 1)
  int n_class;            // classes computed by every workgroup from ONE staged halo (1 forward, 4 data gradient)
  int cls_out;            // classes WRITTEN (= n_class; 1 for the in-place 1x1 data gradient: the other classes keep res)
  int tiles_x, tiles_y;
  int xsc, npix, ny;      // XCD-aware workgroup order, as in Conv2dFwdGeom
  const float *res;       // optional [B][cout][Ho][Wo] added in the epilogue (NULL: none)
  Conv2dStrClass cls[4];
};

// ---------------------------------------------------------------------------------------------------------------------
// The four passes with the tap list known at COMPILE time (round 5).  The first form of this kernel (k_conv2d_str, rounds 3-5,
// deleted in round 6: its bits are in tests/golden/kernel_bits.json) walked run-time class / tap loops: its weight operands are requested right in front of the MFMAs that consume them (18 exposed L2 round trips per
// 32-channel chunk with MTW = 2), its barriers are __syncthreads() (vmcnt(0): the next chunk's values land before anyone
// passes, so the prefetch overlaps nothing) and the stride-2 taps of the forward read the 208-byte pixel rows at a lane stride
// of 416 bytes (4 lanes per LDS slot).  Here, per MODE (0 forward 3x3, 1 forward 1x1, 2 data gradient 3x3, 3 data gradient
// 1x1), the taps are a constexpr table: weight and pixel operands of tap t + 1 are in flight during the MFMAs of tap t (the
// next chunk's first tap during the last), barriers order LDS traffic only, the halo is as large as the mode needs, and the
// forward 3x3 stores its staged columns de-interleaved by parity (even columns, then odd ones: a tap reads 16 CONSECUTIVE
// 224-byte slots, conflict-free like the stride-1 kernel).  Same products in the same order: same bits as k_conv2d_str.
// ---------------------------------------------------------------------------------------------------------------------
struct StrTap {
  int cls, oy, ox, wt;
};
template <int MODE>
__device__ constexpr StrTap str_tap(int t) {
  if (MODE == 0) return StrTap{0, t / 3, t % 3, t};
  if (MODE == 2) {      // classes (py, px) = (0,0) (0,1) (1,0) (1,1) with 1, 2, 2, 4 taps; taps iy-major, as the class descriptors
    const int c = t == 0 ? 0 : (t < 3 ? 1 : (t < 5 ? 2 : 3));
    const int py = c >> 1, px = c & 1;
    const int k = t - (c == 0 ? 0 : (c == 1 ? 1 : (c == 2 ? 3 : 5)));
    const int nx = px ? 2 : 1;
    const int iy = k / nx, ix = k % nx;
    const int ky = py ? (iy == 0 ? 0 : 2) : 1, dy = py ? (iy == 0 ? 1 : 0) : 0;
    const int kx = px ? (ix == 0 ? 0 : 2) : 1, dx = px ? (ix == 0 ? 1 : 0) : 0;
    return StrTap{c, dy, dx, 8 - (3 * ky + kx)};
  }
  return StrTap{0, 0, 0, 0};
}

template <int MODE, int MTW, int OCC, bool FULLA = (MTW == 1)>
__global__ __launch_bounds__(256, OCC) void k_conv2d_str2(const float *__restrict__ in, const unsigned short *__restrict__ Ws,
                                                          Conv2dStrGeom gm, float *__restrict__ out) {
  constexpr int TR = 4, NTW = TR / 2;
  constexpr int S_IN = MODE == 0 ? 2 : 1, NCLS = MODE >= 2 ? 4 : 1, NT = (MODE == 0 || MODE == 2) ? 9 : 1;
  constexpr int EXT = MODE == 0 ? 3 : (MODE == 2 ? 2 : 1);          // halo extent past the last sampled pixel
  constexpr int HR = S_IN * (TR - 1) + EXT, HC = S_IN * 15 + EXT, NPX = HR * HC;
  constexpr int HC0 = (HC + 1) / 2;                                  // MODE 0: even columns first (HC0 of them), then the odd ones
  __shared__ __attribute__((aligned(16))) unsigned char lds[NPX * C2F_PXB1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wid & 1, wn = wid >> 1;
  int bx, by;
  if (!conv2d_xcd_tile(gm.xsc, gm.npix, gm.ny, bx, by)) return;
  const int tx = bx % gm.tiles_x; bx /= gm.tiles_x;
  const int ty = bx % gm.tiles_y;
  const int b = bx / gm.tiles_y;
  const int c0 = tx * 16, r0 = ty * TR;
  const int64_t HWi = (int64_t)gm.Hi * gm.Wi, HWo = (int64_t)gm.Ho * gm.Wo;
  const int n_mt = gm.cout / 16;
  const int mt0 = (by * 2 + wm) * MTW;

  f32x4 acc[NCLS][MTW][NTW];
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int NTASK = (NPX * 4 + 255) / 256;
  int tsrc[NTASK], tdst[NTASK];
  const glb_u8 *const in_b = (const glb_u8 *)in;
#pragma unroll
  for (int r = 0; r < NTASK; ++r) {
    const int task = tid + r * 256;
    const int o = task / NPX, q = task - o * NPX;
    const int qy = q / HC, qx = q - qy * HC;
    const int y = gm.src_stride * (S_IN * r0 + gm.by + qy), x = gm.src_stride * (S_IN * c0 + gm.bx + qx);
    const bool ok = task < NPX * 4 && y >= 0 && y < gm.Hi && x >= 0 && x < gm.Wi;
    tsrc[r] = ok ? (int)((((int64_t)b * gm.cin + 8 * o) * HWi + (int64_t)y * gm.Wi + x) * 4) : -1;
    const int pidx = MODE == 0 ? qy * HC + (qx & 1) * HC0 + (qx >> 1) : q;
    tdst[r] = task < NPX * 4 ? pidx * C2F_PXB1 + o * 16 : -1;
  }
#define C2S_RAW(CH, J, OFF) (*(const glb_f32 *)(c2f_uniform(in_b + ((int64_t)(CH) * 32 + (J)) * HWi * 4) + (unsigned)(OFF)))
  float raw[NTASK][8];
  const int n_chunks = gm.cin / 32;
#pragma unroll
  for (int r = 0; r < NTASK; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[r][j] = tsrc[r] >= 0 ? C2S_RAW(0, j, tsrc[r]) : 0.f;
  const unsigned short *wbase = Ws + (int64_t)mt0 * 3 * 512 + lane * 8;
  // FULLA (one 16-channel block per wave): the operands of ALL taps of a chunk are held in registers and each tap's set is
  // refilled for the NEXT chunk right after its MFMAs -- a prefetch distance of a whole chunk.  One tap ahead (12 MFMAs =
  // 192 cycles) does not cover an L2 round trip at the 2-3 waves per SIMD these kernels run with (forward 256 -> 128 at
  // 96x176: 144 -> 96 us).  Two blocks per wave keep one tap ahead (216 operand registers otherwise), and so does the
  // four-class data gradient of a launch that fills the chip more than once (3 waves per SIMD at 160 registers: 77 vs 93 us).
  constexpr int NA = FULLA ? NT : 1;
  u32x4 ah[NA][MTW], am[NA][MTW], al[NA][MTW], nh[MTW], nm[MTW], nl[MTW];
#define C2S_LOAD_A(CH, WT, H_, M_, L_)                                                                   \
  _Pragma("unroll") for (int mt = 0; mt < MTW; ++mt) {                                                    \
    const unsigned short *wp = wbase + ((((int64_t)(CH) * NT + (WT)) * n_mt + mt) * 3) * 512;            \
    H_[mt] = *(const u32x4 *)(wp);                                                                        \
    M_[mt] = *(const u32x4 *)(wp + 512);                                                                  \
    L_[mt] = *(const u32x4 *)(wp + 1024);                                                                 \
  }
  if (FULLA) {
#pragma unroll
    for (int t = 0; t < NA; ++t) { C2S_LOAD_A(0, str_tap<MODE>(t).wt, ah[t], am[t], al[t]) }
  } else {
    C2S_LOAD_A(0, str_tap<MODE>(0).wt, ah[0], am[0], al[0])
  }
  // pixel operand of tap (oy, ox), tile row nt: MODE 0 reads column 2 li + ox = parity ox & 1, entry li + (ox >> 1)
  const lds_u8 *bbase = (const lds_u8 *)lds + (S_IN * wn * NTW * HC + li) * C2F_PXB1 + g * 16;
#define C2S_LOAD_B(OY, OX, H_, M_, L_)                                                                   \
  _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) {                                                    \
    const lds_u8 *bp = bbase + ((S_IN * nt + (OY)) * HC + (MODE == 0 ? ((OX) & 1) * HC0 + ((OX) >> 1) : (OX))) * C2F_PXB1; \
    H_[nt] = *(const lds_u32x4 *)(bp);                                                                    \
    M_[nt] = *(const lds_u32x4 *)(bp + 64);                                                               \
    L_[nt] = *(const lds_u32x4 *)(bp + 128);                                                              \
  }
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
#pragma unroll
    for (int r = 0; r < NTASK; ++r) {
      if (tdst[r] >= 0) {
        const Split3 s = split_masked(raw[r], 0xffu);
        unsigned char *dst = lds + tdst[r];
        *(u32x4 *)(dst) = s.h;
        *(u32x4 *)(dst + 64) = s.m;
        *(u32x4 *)(dst + 128) = s.l;
      }
    }
    if (chunk + 1 < n_chunks) {
#pragma unroll
      for (int r = 0; r < NTASK; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[r][j] = tsrc[r] >= 0 ? C2S_RAW(chunk + 1, j, tsrc[r]) : 0.f;
    }
    C2F_LDS_BARRIER();
    u32x4 pbh[NTW], pbm[NTW], pbl[NTW];
    C2S_LOAD_B(str_tap<MODE>(0).oy, str_tap<MODE>(0).ox, pbh, pbm, pbl)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const StrTap tp = str_tap<MODE>(t);
      const int ia = FULLA ? t : 0;
      if (!FULLA) {
        if (t + 1 < NT) {
          C2S_LOAD_A(chunk, str_tap<MODE>(t + 1 < NT ? t + 1 : 0).wt, nh, nm, nl)
        } else if (chunk + 1 < n_chunks) {
          C2S_LOAD_A(chunk + 1, str_tap<MODE>(0).wt, nh, nm, nl)
        }
      }
      u32x4 bh[NTW], bm[NTW], bl[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) { bh[nt] = pbh[nt]; bm[nt] = pbm[nt]; bl[nt] = pbl[nt]; }
      if (t + 1 < NT) {
        C2S_LOAD_B(str_tap<MODE>(t + 1 < NT ? t + 1 : 0).oy, str_tap<MODE>(t + 1 < NT ? t + 1 : 0).ox, pbh, pbm, pbl)
      }
      const int c = tp.cls;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        // six products per block, smallest first
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = MFMA_BF16(al[ia][mt], bh[nt], acc[c][mt][nt]);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = MFMA_BF16(am[ia][mt], bm[nt], acc[c][mt][nt]);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = MFMA_BF16(ah[ia][mt], bl[nt], acc[c][mt][nt]);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = MFMA_BF16(am[ia][mt], bh[nt], acc[c][mt][nt]);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = MFMA_BF16(ah[ia][mt], bm[nt], acc[c][mt][nt]);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[c][mt][nt] = MFMA_BF16(ah[ia][mt], bh[nt], acc[c][mt][nt]);
      }
      if (FULLA) {
        if (chunk + 1 < n_chunks) { C2S_LOAD_A(chunk + 1, tp.wt, ah[ia], am[ia], al[ia]) }
      } else {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) { ah[0][mt] = nh[mt]; am[0][mt] = nm[mt]; al[0][mt] = nl[mt]; }
      }
    }
    C2F_LDS_BARRIER();
  }
#undef C2S_RAW
#undef C2S_LOAD_A
#undef C2S_LOAD_B

#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    if (c >= gm.cls_out) break;
    const Conv2dStrClass &cl = gm.cls[c];
    const int col = c0 + li;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = (mt0 + mt) * 16 + 4 * g + j;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int r = r0 + wn * NTW + nt;
          if (col < cl.cols && r < cl.rows) {
            const int64_t o = ((int64_t)b * gm.cout + m) * HWo + (int64_t)(gm.s_out * r + cl.py) * gm.Wo + gm.s_out * col + cl.px;
            out[o] = gm.res ? acc[c][mt][nt][j] + gm.res[o] : acc[c][mt][nt][j];
          }
        }
      }
  }
}

// ksize 3 (padding 1) or 1 (padding 0), stride 2
extern "C" int rslo_conv2d_s2_supported(int cin, int cout, int ksize) {
  return (ksize == 1 || ksize == 3) && cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0;
}

// in [B,cin,H,W] -> out [B,cout,Ho,Wo], Ho = (H - 1) / 2 + 1; Ws = rslo_conv2d_wsplit_k(W, cin, cout, ksize, 0)
extern "C" int rslo_conv2d_fwd_s2(const float *in, const void *Ws, int B, int cin, int cout, int H, int W, int ksize,
                                  float *out, void *stream) {
  RSLO_CHECK_ARG(in && Ws && out && B > 0 && H > 0 && W > 0 && rslo_conv2d_s2_supported(cin, cout, ksize),
                 "rslo_conv2d_fwd_s2: unsupported shape cin=%d cout=%d ksize=%d", cin, cout, ksize);
  Conv2dStrGeom gm = {};
  gm.B = B; gm.cin = cin; gm.cout = cout; gm.Hi = H; gm.Wi = W;
  gm.Ho = (H - 1) / 2 + 1; gm.Wo = (W - 1) / 2 + 1;
  gm.s_out = 1; gm.ntap_w = ksize * ksize; gm.n_class = gm.cls_out = 1;
  gm.tiles_x = (int)rslo_cdiv(gm.Wo, 16); gm.tiles_y = (int)rslo_cdiv(gm.Ho, 4);
  Conv2dStrClass &c = gm.cls[0];
  c.rows = gm.Ho; c.cols = gm.Wo;
  // two 16-channel blocks per wave (64 output channels per workgroup) when the launch still fills the chip: the staged
  // halo and its operand split are shared by twice the MFMAs (256 -> 128 at 96x176: 237 -> 218 us, its 1x1: 41 -> 31 us;
  // the smaller stages lose, 44 -> 62 us, and keep 32 channels)
  const int mtw_env = rslo_tune(RSLO_TUNE_CONV2D_S2_MTW);
  // (k_conv2d_str2 holds a chunk's weight operands in registers with ONE block per wave: 96 vs 161 us on that layer; two are opt-in)
  const int mtw = (cout % 64 == 0 && mtw_env == 2) ? 2 : 1;
  gm.npix = B * gm.tiles_x * gm.tiles_y;
  gm.ny = cout / (32 * mtw);
  gm.xsc = conv2d_xcd_split(RSLO_TUNE_CONV2D_S2_XSC, gm.ny, 6.0 * ksize * ksize * cin * cout, 4.0 * B * cin * H * W * (ksize == 3 ? 1.5 : 0.25));
  const dim3 grid = conv2d_xcd_grid(gm.xsc, gm.npix, gm.ny);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *ws = (const unsigned short *)Ws;
  if (ksize == 3) {
    gm.src_stride = 1; gm.by = gm.bx = -1; c.ny = c.nx = 3;
    for (int i = 0; i < 3; ++i) c.oy[i] = c.ox[i] = i;
    for (int i = 0; i < 9; ++i) c.wt[i] = i;
    if (mtw == 2) hipLaunchKernelGGL((k_conv2d_str2<0, 2, 2>), grid, dim3(256), 0, st, in, ws, gm, out);
    else hipLaunchKernelGGL((k_conv2d_str2<0, 1, 2>), grid, dim3(256), 0, st, in, ws, gm, out);
  } else {        // out[y][x] = W in[2y][2x]: stage only the sampled pixels
    gm.src_stride = 2; c.ny = c.nx = 1;
    if (mtw == 2) hipLaunchKernelGGL((k_conv2d_str2<1, 2, 4>), grid, dim3(256), 0, st, in, ws, gm, out);
    else hipLaunchKernelGGL((k_conv2d_str2<1, 1, 4>), grid, dim3(256), 0, st, in, ws, gm, out);
  }
  RSLO_CHECK_LAUNCH("k_conv2d_str2(fwd)");
  return RSLO_OK;
}

// dout [B,cout,Ho,Wo] -> din [B,cin,H,W] (every element written); Ws = rslo_conv2d_wsplit_k(W, cin, cout, ksize, 1)
extern "C" int rslo_conv2d_dgrad_s2_add(const float *dout, const void *Ws, const float *res, int B, int cin, int cout, int H,
                                        int W, int ksize, float *din, void *stream);
extern "C" int rslo_conv2d_dgrad_s2(const float *dout, const void *Ws, int B, int cin, int cout, int H, int W, int ksize,
                                    float *din, void *stream) {
  return rslo_conv2d_dgrad_s2_add(dout, Ws, nullptr, B, cin, cout, H, W, ksize, din, stream);
}

// din = data gradient + res (res [B,cin,H,W], not aliasing din): the two input gradients of a stride-2 BasicBlock
// (3x3 branch + 1x1 downsample branch) meet in the second one's epilogue
extern "C" int rslo_conv2d_dgrad_s2_add(const float *dout, const void *Ws, const float *res, int B, int cin, int cout, int H,
                                        int W, int ksize, float *din, void *stream) {
  RSLO_CHECK_ARG(dout && Ws && din && B > 0 && H > 0 && W > 0 && rslo_conv2d_s2_supported(cin, cout, ksize),
                 "rslo_conv2d_dgrad_s2: unsupported shape cin=%d cout=%d ksize=%d", cin, cout, ksize);
  Conv2dStrGeom gm = {};
  gm.B = B; gm.cin = cout; gm.cout = cin;       // contraction over the forward's output channels
  gm.Hi = (H - 1) / 2 + 1; gm.Wi = (W - 1) / 2 + 1; gm.Ho = H; gm.Wo = W;
  gm.s_out = 2; gm.ntap_w = ksize * ksize; gm.n_class = 4; gm.src_stride = 1;
  gm.res = res;
  // 1x1 in place (res == din): the gradient lands on the pixels (2y, 2x) only, every other element of din keeps the value it
  // has -- a quarter of the elements is read and written instead of a full read of res and a full write of din
  RSLO_CHECK_ARG(res != din || ksize == 1, "rslo_conv2d_dgrad_s2_add: res may alias din for ksize 1 only");
  gm.cls_out = (res == din && res != nullptr) ? 1 : 4;
  const int rmax = (H + 1) / 2, cmax = (W + 1) / 2;
  gm.tiles_x = (int)rslo_cdiv(cmax, 16); gm.tiles_y = (int)rslo_cdiv(rmax, 4);
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      Conv2dStrClass &c = gm.cls[py * 2 + px];
      c.py = py; c.px = px;
      c.rows = (H - py + 1) / 2; c.cols = (W - px + 1) / 2;
      if (ksize == 3) {
        // Y = 2y + ky - 1 = 2r + py  ->  py = 0: ky = 1, y = r;  py = 1: ky = 0, y = r + 1 and ky = 2, y = r.  The operand
        // with transpose = 1 stores original tap t at index 8 - t.
        int kys[2], dys[2], kxs[2], dxs[2];
        c.ny = py ? 2 : 1; c.nx = px ? 2 : 1;
        if (py) { kys[0] = 0; dys[0] = 1; kys[1] = 2; dys[1] = 0; } else { kys[0] = 1; dys[0] = 0; }
        if (px) { kxs[0] = 0; dxs[0] = 1; kxs[1] = 2; dxs[1] = 0; } else { kxs[0] = 1; dxs[0] = 0; }
        for (int iy = 0; iy < c.ny; ++iy) c.oy[iy] = dys[iy];
        for (int ix = 0; ix < c.nx; ++ix) c.ox[ix] = dxs[ix];
        for (int iy = 0; iy < c.ny; ++iy)
          for (int ix = 0; ix < c.nx; ++ix) c.wt[iy * 3 + ix] = 8 - (3 * kys[iy] + kxs[ix]);
      } else {
        c.ny = c.nx = (py == 0 && px == 0) ? 1 : 0;      // din[2y][2x] only; the other classes are zeros
      }
    }
  // all four classes of a tile from one staged halo of dout (9 tap products in total for 3x3)
  const int mtw_env = rslo_tune(RSLO_TUNE_CONV2D_S2_MTW);
  const int64_t wgs32 = (int64_t)B * gm.tiles_x * gm.tiles_y * (cin / 32);
  // measured: 64 channels per workgroup loses on the four-class data gradient (138 vs 100 us on the
                    // largest layer: 4 x 2 x 2 accumulator tiles per wave), so it is opt-in here
  const int mtw = (cin % 64 == 0 && mtw_env == 2) ? 2 : 1;
  gm.npix = B * gm.tiles_x * gm.tiles_y;
  gm.ny = cin / (32 * mtw);
  gm.xsc = conv2d_xcd_split(RSLO_TUNE_CONV2D_S2_XSC, gm.ny, 6.0 * ksize * ksize * cin * cout, 4.0 * B * cout * gm.Hi * gm.Wi * 1.5);
  const dim3 grid = conv2d_xcd_grid(gm.xsc, gm.npix, gm.ny);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *ws = (const unsigned short *)Ws;
  if (ksize == 3 && mtw == 2) hipLaunchKernelGGL((k_conv2d_str2<2, 2, 2>), grid, dim3(256), 0, st, dout, ws, gm, din);
  else if (ksize == 3 && wgs32 > 512) hipLaunchKernelGGL((k_conv2d_str2<2, 1, 3, false>), grid, dim3(256), 0, st, dout, ws, gm, din);
  else if (ksize == 3) hipLaunchKernelGGL((k_conv2d_str2<2, 1, 2, true>), grid, dim3(256), 0, st, dout, ws, gm, din);
  else if (mtw == 2) hipLaunchKernelGGL((k_conv2d_str2<3, 2, 2>), grid, dim3(256), 0, st, dout, ws, gm, din);
  else hipLaunchKernelGGL((k_conv2d_str2<3, 1, 3>), grid, dim3(256), 0, st, dout, ws, gm, din);
  RSLO_CHECK_LAUNCH("k_conv2d_str2(dgrad)");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// 1x1 output convolutions of the head (32/64 -> 7 or 1 channels: tq_map_conv.6, pyramid_motion_blocks.*.6, the
// confidence logits): HBM-bound row operations, one thread per pixel.  The library path spends 5 launches per backward
// on them (igemm + layout transposes + fills); here forward, data gradient and weight + bias gradient are one or two.
// ---------------------------------------------------------------------------------------------------------------------
#define C11_MAXCO 8
#define C11_MAXW (C11_MAXCO * 256)

// out[b][co][p] = bias[co] + sum_ci W[co][ci] x[b][ci][p]
__global__ __launch_bounds__(256) void k_conv1x1_fwd(const float *__restrict__ x, const float *__restrict__ W,
                                                     const float *__restrict__ bias, int B, int cin, int cout, int HW,
                                                     float *__restrict__ out) {
  __shared__ float w[C11_MAXW];
  for (int e = threadIdx.x; e < cout * cin; e += 256) w[e] = W[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * HW) return;
  const int b = (int)(i / HW), p = (int)(i - (int64_t)b * HW);
  float acc[C11_MAXCO];
#pragma unroll
  for (int co = 0; co < C11_MAXCO; ++co) acc[co] = (bias && co < cout) ? bias[co] : 0.f;
  const float *xp = x + (int64_t)b * cin * HW + p;
  int ci = 0;
  for (; ci + 8 <= cin; ci += 8) {      // 8 channel loads in flight (a serial chain of cin loads was 34 us on a 24x44 map),
    float xv[8];                         // accumulated in channel order: same bits
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = xp[(int64_t)(ci + u) * HW];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int co = 0; co < C11_MAXCO; ++co)
        if (co < cout) acc[co] += w[co * cin + ci + u] * xv[u];
  }
  for (; ci < cin; ++ci) {
    const float xv = xp[(int64_t)ci * HW];
#pragma unroll
    for (int co = 0; co < C11_MAXCO; ++co)
      if (co < cout) acc[co] += w[co * cin + ci] * xv;
  }
  float *op = out + (int64_t)b * cout * HW + p;
#pragma unroll
  for (int co = 0; co < C11_MAXCO; ++co)
    if (co < cout) op[(int64_t)co * HW] = acc[co];
}

// dx[b][ci][p] = sum_co W[co][ci] dy[b][co][p]
__global__ __launch_bounds__(256) void k_conv1x1_dgrad(const float *__restrict__ dy, const float *__restrict__ W, int B,
                                                       int cin, int cout, int HW, float *__restrict__ dx) {
  __shared__ float w[C11_MAXW];
  for (int e = threadIdx.x; e < cout * cin; e += 256) w[e] = W[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * HW) return;
  const int b = (int)(i / HW), p = (int)(i - (int64_t)b * HW);
  float g[C11_MAXCO];
  const float *gp = dy + (int64_t)b * cout * HW + p;
#pragma unroll
  for (int co = 0; co < C11_MAXCO; ++co) g[co] = co < cout ? gp[(int64_t)co * HW] : 0.f;
  float *dp = dx + (int64_t)b * cin * HW + p;
  for (int ci = 0; ci < cin; ++ci) {
    float s = 0.f;
#pragma unroll
    for (int co = 0; co < C11_MAXCO; ++co)
      if (co < cout) s += w[co * cin + ci] * g[co];
    dp[(int64_t)ci * HW] = s;
  }
}

// The same two passes with the CHANNELS of a pixel dealt to the four waves of a workgroup (cin <= 64; round 5).  One thread
// per pixel is 1056 waves on the 96x176 maps -- one per SIMD -- and its inner loops were latency chains: the weights sat in LDS
// behind a run-time `co < cout` test, so every product waited for its own ds_read (0.3 us per channel: 20 us even on a 24x44
// map).  Here COUT is a template parameter and the weights are read straight from global memory at wave-uniform addresses
// (scalar loads into SGPRs, no LDS); a workgroup takes 64 pixels, wave w loads channels [w G, (w + 1) G) of them at once
// (G = ceil(cin / 4) <= 16 loads in flight per lane), and the waves add their products onto the running sums one after the
// other through LDS: the additions of a pixel happen in channel order exactly as in k_conv1x1_fwd -- same bits.
template <int COUT>
__global__ __launch_bounds__(256) void k_conv1x1_fwd4(const float *__restrict__ x, const float *__restrict__ W,
                                                      const float *__restrict__ bias, int B, int cin, int HW,
                                                      float *__restrict__ out) {
  __shared__ float run[COUT][64];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = i < (int64_t)B * HW;
  const int b = valid ? (int)(i / HW) : 0, p = valid ? (int)(i - (int64_t)b * HW) : 0;
  const int G = (cin + 3) >> 2, c0 = wid * G, c1 = (c0 + G < cin) ? c0 + G : cin;
  const float *xp = x + (int64_t)b * cin * HW + p;
  float xv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) xv[u] = (valid && c0 + u < c1) ? xp[(int64_t)(c0 + u) * HW] : 0.f;
  float acc[COUT];
  for (int step = 0; step < 4; ++step) {
    if (wid == step) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = step == 0 ? (bias ? bias[co] : 0.f) : run[co][lane];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (c0 + u < c1) {
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] += W[co * cin + c0 + u] * xv[u];
        }
      if (step < 3) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) run[co][lane] = acc[co];
      } else if (valid) {
        float *op = out + (int64_t)b * COUT * HW + p;
#pragma unroll
        for (int co = 0; co < COUT; ++co) op[(int64_t)co * HW] = acc[co];
      }
    }
    __syncthreads();
  }
}

// dx channels [w G, (w + 1) G) of 64 pixels per wave: every channel's sum is formed as in k_conv1x1_dgrad (same bits)
template <int COUT>
__global__ __launch_bounds__(256) void k_conv1x1_dgrad4(const float *__restrict__ dy, const float *__restrict__ W, int B,
                                                        int cin, int HW, float *__restrict__ dx) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  if (i >= (int64_t)B * HW) return;
  const int b = (int)(i / HW), p = (int)(i - (int64_t)b * HW);
  float g[COUT];
  const float *gp = dy + (int64_t)b * COUT * HW + p;
#pragma unroll
  for (int co = 0; co < COUT; ++co) g[co] = gp[(int64_t)co * HW];
  const int G = (cin + 3) >> 2, c0 = wid * G, c1 = (c0 + G < cin) ? c0 + G : cin;
  float *dp = dx + (int64_t)b * cin * HW + p;
  for (int ci = c0; ci < c1; ++ci) {
    float s = 0.f;
#pragma unroll
    for (int co = 0; co < COUT; ++co) s += W[co * cin + ci] * g[co];
    dp[(int64_t)ci * HW] = s;
  }
}

// grid (cin + 1, S): block (ci, slab) sums dy[b][co][p] * x[b][ci][p] over its pixel slab (ci == cin: x = 1, the bias
// gradient); part [S][cin + 1][C11_MAXCO]
__global__ __launch_bounds__(256) void k_conv1x1_wgrad(const float *__restrict__ x, const float *__restrict__ dy, int B,
                                                       int cin, int cout, int HW, int64_t per_slab,
                                                       float *__restrict__ part) {
  const int ci = blockIdx.x, slab = blockIdx.y;
  const int64_t total = (int64_t)B * HW;
  const int64_t i0 = (int64_t)slab * per_slab, i1 = (i0 + per_slab < total) ? i0 + per_slab : total;
  float acc[C11_MAXCO];
#pragma unroll
  for (int co = 0; co < C11_MAXCO; ++co) acc[co] = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const int b = (int)(i / HW), p = (int)(i - (int64_t)b * HW);
    const float xv = ci < cin ? x[((int64_t)b * cin + ci) * HW + p] : 1.f;
    const float *gp = dy + (int64_t)b * cout * HW + p;
#pragma unroll
    for (int co = 0; co < C11_MAXCO; ++co)
      if (co < cout) acc[co] += gp[(int64_t)co * HW] * xv;
  }
  __shared__ float red[C11_MAXCO][4];
#pragma unroll
  for (int co = 0; co < C11_MAXCO; ++co) {
    float v = acc[co];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[co][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < C11_MAXCO)
    part[((int64_t)slab * (cin + 1) + ci) * C11_MAXCO + threadIdx.x] =
        ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
}

// dW[co][ci] (and dbias[co] from the virtual channel ci == cin) = slab partials added in slab order
__global__ void k_conv1x1_wgrad_reduce(const float *__restrict__ part, int S, int cin, int cout, float *__restrict__ dW,
                                       float *__restrict__ dbias) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (cin + 1) * cout) return;
  const int ci = e / cout, co = e - ci * cout;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[((int64_t)k * (cin + 1) + ci) * C11_MAXCO + co];
  if (ci < cin) dW[co * cin + ci] = s;
  else if (dbias) dbias[co] = s;
}

static int conv1x1_slabs(int B, int cin, int HW, int64_t *per) {
  const int64_t total = (int64_t)B * HW;
  int s = 1024 / (cin + 1);
  if (s < 1) s = 1;
  const int64_t max_s = rslo_cdiv(total, 2048);
  if (s > max_s) s = (int)(max_s > 0 ? max_s : 1);
  *per = rslo_cdiv(total, s);
  return (int)rslo_cdiv(total, *per);
}

extern "C" int rslo_conv1x1_supported(int cin, int cout) { return cin >= 1 && cin <= 256 && cout >= 1 && cout <= C11_MAXCO; }

extern "C" size_t rslo_conv1x1_wgrad_ws_bytes(int B, int cin, int cout, int HW) {
  int64_t per;
  return (size_t)conv1x1_slabs(B, cin, HW, &per) * (cin + 1) * C11_MAXCO * sizeof(float);
}

extern "C" int rslo_conv1x1_fwd(const float *x, const float *W, const float *bias, int B, int cin, int cout, int HW,
                                float *out, void *stream) {
  RSLO_CHECK_ARG(x && W && out && rslo_conv1x1_supported(cin, cout) && B >= 1 && HW >= 1, "rslo_conv1x1_fwd: bad arguments");
  if (cin <= 64 && rslo_tune(RSLO_TUNE_CONV1X1_SPLIT)) {       // channels dealt to the four waves of a workgroup
    const dim3 grid4((unsigned)rslo_cdiv((int64_t)B * HW, 64));
#define C11_F4(N) case N: hipLaunchKernelGGL((k_conv1x1_fwd4<N>), grid4, dim3(256), 0, (hipStream_t)stream, x, W, bias, B, cin, HW, out); break;
    switch (cout) { C11_F4(1) C11_F4(2) C11_F4(3) C11_F4(4) C11_F4(5) C11_F4(6) C11_F4(7) C11_F4(8) }
#undef C11_F4
  } else
    hipLaunchKernelGGL(k_conv1x1_fwd, dim3((unsigned)rslo_cdiv((int64_t)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       W, bias, B, cin, cout, HW, out);
  RSLO_CHECK_LAUNCH("k_conv1x1_fwd");
  return RSLO_OK;
}

extern "C" int rslo_conv1x1_dgrad(const float *dy, const float *W, int B, int cin, int cout, int HW, float *dx,
                                  void *stream) {
  RSLO_CHECK_ARG(dy && W && dx && rslo_conv1x1_supported(cin, cout) && B >= 1 && HW >= 1, "rslo_conv1x1_dgrad: bad arguments");
  if (cin <= 64 && rslo_tune(RSLO_TUNE_CONV1X1_SPLIT)) {
    const dim3 grid4((unsigned)rslo_cdiv((int64_t)B * HW, 64));
#define C11_D4(N) case N: hipLaunchKernelGGL((k_conv1x1_dgrad4<N>), grid4, dim3(256), 0, (hipStream_t)stream, dy, W, B, cin, HW, dx); break;
    switch (cout) { C11_D4(1) C11_D4(2) C11_D4(3) C11_D4(4) C11_D4(5) C11_D4(6) C11_D4(7) C11_D4(8) }
#undef C11_D4
  } else
    hipLaunchKernelGGL(k_conv1x1_dgrad, dim3((unsigned)rslo_cdiv((int64_t)B * HW, 256)), dim3(256), 0, (hipStream_t)stream,
                       dy, W, B, cin, cout, HW, dx);
  RSLO_CHECK_LAUNCH("k_conv1x1_dgrad");
  return RSLO_OK;
}

extern "C" int rslo_conv1x1_wgrad(const float *x, const float *dy, int B, int cin, int cout, int HW, float *dW,
                                  float *dbias, void *ws, size_t ws_bytes, void *stream) {
  RSLO_CHECK_ARG(x && dy && dW && ws && rslo_conv1x1_supported(cin, cout) && B >= 1 && HW >= 1, "rslo_conv1x1_wgrad: bad arguments");
  RSLO_CHECK_ARG(ws_bytes >= rslo_conv1x1_wgrad_ws_bytes(B, cin, cout, HW), "rslo_conv1x1_wgrad: workspace too small");
  int64_t per;
  const int S = conv1x1_slabs(B, cin, HW, &per);
  hipLaunchKernelGGL(k_conv1x1_wgrad, dim3((unsigned)(cin + 1), (unsigned)S), dim3(256), 0, (hipStream_t)stream, x, dy, B,
                     cin, cout, HW, per, (float *)ws);
  RSLO_CHECK_LAUNCH("k_conv1x1_wgrad");
  hipLaunchKernelGGL(k_conv1x1_wgrad_reduce, dim3((unsigned)rslo_cdiv((cin + 1) * cout, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float *)ws, S, cin, cout, dW, dbias);
  RSLO_CHECK_LAUNCH("k_conv1x1_wgrad_reduce");
  return RSLO_OK;
}
