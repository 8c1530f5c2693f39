// The element-wise tail of the BEV head between its last convolutions and the vote / the loss
// (reference: rslo/models/odom_pred.py:227-264 with rslo/layers/confidence.py:26-34):
//
//   tq_map   = cat(t, q / |q|)                                        quaternion channels normalised per cell
//   conf     = softmax over the H*W cells of where(mask > 0, logit, -1000) / T       for both confidence heads, T = 1
//              (carries the gradient) and T = 20 (loss masks, no gradient)
//   masks    = occupancy pyramid (MaxPool 3/2/1), loss weights w0 = mask * conf_T20, w_{k+1} = m_{k+1} * AvgPool(3/2/1)(w_k),
//              masked pyramid predictions p * (m > 0), masked pose maps
//
// As torch ops this is ~45 launches forward and ~45 backward per step on maps of a few 10^4 cells -- each one costs more
// in launch gap than in work.  Five small launches forward (three for the mask pyramid's levels), three backward.
#include "rslo_common.h"

// ----------------------------------------------------------------------------------------- quaternion normalisation
__global__ void k_tq_normalize_fwd(const float *__restrict__ in, int64_t cells, float *__restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  const int64_t base = (int64_t)blockIdx.y * 7 * cells + c;
#pragma unroll
  for (int k = 0; k < 3; ++k) out[base + k * cells] = in[base + k * cells];
  float q[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = in[base + (3 + k) * cells];
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);      // torch.norm: no epsilon
#pragma unroll
  for (int k = 0; k < 4; ++k) out[base + (3 + k) * cells] = q[k] / n;
}

__global__ void k_tq_normalize_bwd(const float *__restrict__ in, const float *__restrict__ g, int64_t cells,
                                   float *__restrict__ din) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  const int64_t base = (int64_t)blockIdx.y * 7 * cells + c;
#pragma unroll
  for (int k = 0; k < 3; ++k) din[base + k * cells] = g[base + k * cells];
  float q[4], gq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    q[k] = in[base + (3 + k) * cells];
    gq[k] = g[base + (3 + k) * cells];
  }
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float dot = (q[0] * gq[0] + q[1] * gq[1] + q[2] * gq[2] + q[3] * gq[3]) / (n * n);      // q_hat . g / n
#pragma unroll
  for (int k = 0; k < 4; ++k) din[base + (3 + k) * cells] = gq[k] / n - q[k] * dot / n;
}

extern "C" int rslo_tq_normalize_fwd(const float *tq, int B, int64_t cells, float *out, void *stream) {
  RSLO_CHECK_ARG(tq && out && B >= 1 && B < 65536 && cells >= 1, "rslo_tq_normalize_fwd: bad arguments");
  hipLaunchKernelGGL(k_tq_normalize_fwd, dim3((unsigned)rslo_cdiv(cells, 256), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, tq, cells, out);
  RSLO_CHECK_LAUNCH("k_tq_normalize_fwd");
  return RSLO_OK;
}

extern "C" int rslo_tq_normalize_bwd(const float *tq, const float *grad, int B, int64_t cells, float *dtq, void *stream) {
  RSLO_CHECK_ARG(tq && grad && dtq && B >= 1 && B < 65536 && cells >= 1, "rslo_tq_normalize_bwd: bad arguments");
  hipLaunchKernelGGL(k_tq_normalize_bwd, dim3((unsigned)rslo_cdiv(cells, 256), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, tq, grad, cells, dtq);
  RSLO_CHECK_LAUNCH("k_tq_normalize_bwd");
  return RSLO_OK;
}

// ----------------------------------------------------------------------------------------- masked spatial softmax
#define CS_THREADS 1024
#define CS_MAXPER 32          // cells per thread held in registers: H * W <= 32768

__device__ __forceinline__ float cs_block_reduce(float v, bool is_max, float *sm) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float o = __shfl_xor(v, d, 64);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();             // sm may still be read from the previous reduction
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  float r = sm[0];
  for (int w = 1; w < CS_THREADS / 64; ++w) r = is_max ? fmaxf(r, sm[w]) : r + sm[w];
  return r;
}

// grid (B, 2): block (b, h) handles head h (0: translation, 1: rotation) of sample b.
// conf1 [2][B, cells] (T = 1), confT [B, 2, cells] (temperature `temp`, the layout of torch.cat([t, r], 1)).
__global__ __launch_bounds__(CS_THREADS) void k_conf_softmax_fwd(const float *__restrict__ t_logit,
                                                                 const float *__restrict__ r_logit,
                                                                 const unsigned char *__restrict__ outside, int cells,
                                                                 float temp, float *__restrict__ t_conf,
                                                                 float *__restrict__ r_conf, float *__restrict__ confT) {
  __shared__ float sm[CS_THREADS / 64];
  const int b = blockIdx.x, h = blockIdx.y, B = gridDim.x;
  const float *lg = (h ? r_logit : t_logit) + (int64_t)b * cells;
  const unsigned char *om = outside + (int64_t)b * cells;
  float z[CS_MAXPER];
  float m1 = -INFINITY;
#pragma unroll
  for (int j = 0; j < CS_MAXPER; ++j) {
    const int i = threadIdx.x + j * CS_THREADS;
    z[j] = i < cells ? (om[i] ? -1000.0f : lg[i]) : -INFINITY;
    m1 = fmaxf(m1, z[j]);
  }
  m1 = cs_block_reduce(m1, true, sm);
  float s1 = 0.f, sT = 0.f;
  const float mT = m1 / temp;
#pragma unroll
  for (int j = 0; j < CS_MAXPER; ++j) {
    const int i = threadIdx.x + j * CS_THREADS;
    if (i < cells) {
      s1 += expf(z[j] - m1);
      sT += expf(z[j] / temp - mT);
    }
  }
  s1 = cs_block_reduce(s1, false, sm);
  sT = cs_block_reduce(sT, false, sm);
  float *o1 = (h ? r_conf : t_conf) + (int64_t)b * cells;
  float *oT = confT + ((int64_t)b * 2 + h) * cells;
  (void)B;
#pragma unroll
  for (int j = 0; j < CS_MAXPER; ++j) {
    const int i = threadIdx.x + j * CS_THREADS;
    if (i < cells) {
      o1[i] = expf(z[j] - m1) / s1;
      oT[i] = expf(z[j] / temp - mT) / sT;
    }
  }
}

// d logit = outside ? 0 : conf (g - sum_j conf_j g_j)
__global__ __launch_bounds__(CS_THREADS) void k_conf_softmax_bwd(const float *__restrict__ t_conf,
                                                                 const float *__restrict__ r_conf,
                                                                 const float *__restrict__ g_t,
                                                                 const float *__restrict__ g_r,
                                                                 const unsigned char *__restrict__ outside, int cells,
                                                                 float *__restrict__ d_t, float *__restrict__ d_r) {
  __shared__ float sm[CS_THREADS / 64];
  const int b = blockIdx.x, h = blockIdx.y;
  const float *cf = (h ? r_conf : t_conf) + (int64_t)b * cells;
  const float *g = (h ? g_r : g_t) + (int64_t)b * cells;
  const unsigned char *om = outside + (int64_t)b * cells;
  float *d = (h ? d_r : d_t) + (int64_t)b * cells;
  float c[CS_MAXPER], gg[CS_MAXPER];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < CS_MAXPER; ++j) {
    const int i = threadIdx.x + j * CS_THREADS;
    c[j] = i < cells ? cf[i] : 0.f;
    gg[j] = i < cells ? g[i] : 0.f;
    dot += c[j] * gg[j];
  }
  dot = cs_block_reduce(dot, false, sm);
#pragma unroll
  for (int j = 0; j < CS_MAXPER; ++j) {
    const int i = threadIdx.x + j * CS_THREADS;
    if (i < cells) d[i] = om[i] ? 0.f : c[j] * (gg[j] - dot);
  }
}

extern "C" int rslo_conf_softmax_fwd(const float *t_logit, const float *r_logit, const unsigned char *outside, int B,
                                     int cells, float temperature, float *t_conf, float *r_conf, float *conf_temp,
                                     void *stream) {
  RSLO_CHECK_ARG(t_logit && r_logit && outside && t_conf && r_conf && conf_temp, "rslo_conf_softmax_fwd: null pointer");
  RSLO_CHECK_ARG(B >= 1 && B < 65536 && cells >= 1 && cells <= CS_THREADS * CS_MAXPER && temperature > 0.f,
                 "rslo_conf_softmax_fwd: B / cells out of range (cells <= 32768)");
  hipLaunchKernelGGL(k_conf_softmax_fwd, dim3((unsigned)B, 2), dim3(CS_THREADS), 0, (hipStream_t)stream, t_logit, r_logit,
                     outside, cells, temperature, t_conf, r_conf, conf_temp);
  RSLO_CHECK_LAUNCH("k_conf_softmax_fwd");
  return RSLO_OK;
}

extern "C" int rslo_conf_softmax_bwd(const float *t_conf, const float *r_conf, const float *g_t, const float *g_r,
                                     const unsigned char *outside, int B, int cells, float *d_t_logit, float *d_r_logit,
                                     void *stream) {
  RSLO_CHECK_ARG(t_conf && r_conf && g_t && g_r && outside && d_t_logit && d_r_logit, "rslo_conf_softmax_bwd: null pointer");
  RSLO_CHECK_ARG(B >= 1 && B < 65536 && cells >= 1 && cells <= CS_THREADS * CS_MAXPER, "rslo_conf_softmax_bwd: bad sizes");
  hipLaunchKernelGGL(k_conf_softmax_bwd, dim3((unsigned)B, 2), dim3(CS_THREADS), 0, (hipStream_t)stream, t_conf, r_conf,
                     g_t, g_r, outside, cells, d_t_logit, d_r_logit);
  RSLO_CHECK_LAUNCH("k_conf_softmax_bwd");
  return RSLO_OK;
}

// ----------------------------------------------------------------------------------------- mask / weight pyramid
// Level 0 = the H x W map; level k + 1 = level k pooled with kernel 3, stride 2, padding 1 (H, W even: half the size).
//   occ_0 = mask;                 occ_{k+1} = MaxPool(occ_k)            (padding ignored by the max)
//   w_0   = mask * conf_T [2 ch]; w_{k+1}   = occ_{k+1} * AvgPool(w_k)  (zero padding counted: divisor 9)
// One small launch per level (a coarser level reads the one below; recomputing everything from level 0 in ONE launch left
// a few blocks walking 7 x 7 windows: 70 us).  Masked maps: pred_k * (occ_k > 0) for the pyramid predictions (k >= 1),
// tq * mask and tq_g * mask at level 0.
struct HeadMaskArgs {
  const float *mask;        // [B, 1, H, W] float 0/1
  const float *conf;        // [B, 2, H, W]
  const float *tq, *tq_g;   // [B, 7, H, W]
  const float *pred[3];     // pred[k-1]: [B, 7, H >> k, W >> k], k = 1 .. levels - 1
  float *w[4];              // w[k]: [B, 2, H >> k, W >> k]
  float *occ[4];            // occ[k], k >= 1: [B, 1, H >> k, W >> k]
  float *mpred[3];          // masked predictions
  float *mtq, *mtq_g;
  int B, H, W, levels;      // levels = 1 + number of pyramid predictions (<= 4)
};

// one launch per level, coarser levels read the level below (stream order): grid (cells of level k / 256, B)
__global__ void k_head_masks_fwd(HeadMaskArgs a, int k) {
  const int b = blockIdx.y;
  const int h = a.H >> k, w = a.W >> k;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= h * w) return;
  const int64_t cells = (int64_t)h * w;
  if (k == 0) {
    const float m = a.mask[(int64_t)b * cells + c];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) a.w[0][((int64_t)b * 2 + ch) * cells + c] = m * a.conf[((int64_t)b * 2 + ch) * cells + c];
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) {
      const int64_t i = ((int64_t)b * 7 + ch) * cells + c;
      a.mtq[i] = a.tq[i] * m;
      a.mtq_g[i] = a.tq_g[i] * m;
    }
    return;
  }
  const int y = c / w, x = c - y * w;
  const int hp = a.H >> (k - 1), wp = a.W >> (k - 1);
  const float *occ_p = (k == 1 ? a.mask : a.occ[k - 1]) + (int64_t)b * hp * wp;
  const float *w_p = a.w[k - 1] + (int64_t)b * 2 * hp * wp;
  float o = -INFINITY, s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = 2 * y + dy, xx = 2 * x + dx;
      if (yy < 0 || xx < 0 || yy >= hp || xx >= wp) continue;      // MaxPool ignores the padding, AvgPool adds zeros
      o = fmaxf(o, occ_p[yy * wp + xx]);
      s0 += w_p[yy * wp + xx];
      s1 += w_p[(int64_t)hp * wp + yy * wp + xx];
    }
  a.occ[k][(int64_t)b * cells + c] = o;
  a.w[k][((int64_t)b * 2 + 0) * cells + c] = o * (s0 / 9.0f);
  a.w[k][((int64_t)b * 2 + 1) * cells + c] = o * (s1 / 9.0f);
  const float keep = o > 0.f ? 1.f : 0.f;
#pragma unroll
  for (int ch = 0; ch < 7; ++ch) {
    const int64_t i = ((int64_t)b * 7 + ch) * cells + c;
    a.mpred[k - 1][i] = a.pred[k - 1][i] * keep;
  }
}

struct HeadMaskBwdArgs {
  const float *mask;          // level-0 mask
  const float *occ[4];        // occ[k], k >= 1
  const float *g_mpred[3];    // gradients of the masked predictions (may be NULL: no gradient arrived)
  const float *g_mtq;         // may be NULL
  float *d_pred[3];
  float *d_tq;
  int B, H, W, levels;
};

__global__ void k_head_masks_bwd(HeadMaskBwdArgs a) {
  const int k = blockIdx.z, b = blockIdx.y;
  const int h = a.H >> k, w = a.W >> k;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= h * w) return;
  const int64_t cells = (int64_t)h * w;
  if (k == 0) {
    const float m = a.mask[(int64_t)b * cells + c];
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) {
      const int64_t i = ((int64_t)b * 7 + ch) * cells + c;
      a.d_tq[i] = a.g_mtq ? a.g_mtq[i] * m : 0.f;
    }
    return;
  }
  const float keep = a.occ[k][(int64_t)b * cells + c] > 0.f ? 1.f : 0.f;
#pragma unroll
  for (int ch = 0; ch < 7; ++ch) {
    const int64_t i = ((int64_t)b * 7 + ch) * cells + c;
    a.d_pred[k - 1][i] = a.g_mpred[k - 1] ? a.g_mpred[k - 1][i] * keep : 0.f;
  }
}

extern "C" int rslo_head_masks_fwd(const RsloHeadMasks *h_a, void *stream) {
  RSLO_CHECK_ARG(h_a && h_a->B >= 1 && h_a->B < 65536 && h_a->levels >= 1 && h_a->levels <= 4, "rslo_head_masks_fwd: bad sizes");
  RSLO_CHECK_ARG(h_a->H % (1 << (h_a->levels - 1)) == 0 && h_a->W % (1 << (h_a->levels - 1)) == 0,
                 "rslo_head_masks_fwd: H, W must be divisible by 2^(levels-1)");
  HeadMaskArgs a;
  a.mask = h_a->mask; a.conf = h_a->conf; a.tq = h_a->tq; a.tq_g = h_a->tq_g; a.mtq = h_a->mtq; a.mtq_g = h_a->mtq_g;
  a.B = h_a->B; a.H = h_a->H; a.W = h_a->W; a.levels = h_a->levels;
  RSLO_CHECK_ARG(a.mask && a.conf && a.tq && a.tq_g && a.mtq && a.mtq_g && h_a->w[0], "rslo_head_masks_fwd: null pointer");
  for (int k = 0; k < 4; ++k) {
    a.w[k] = h_a->w[k];
    a.occ[k] = h_a->occ[k];
    if (k < 3) {
      a.pred[k] = h_a->pred[k];
      a.mpred[k] = h_a->mpred[k];
    }
    if (k >= 1 && k < a.levels)
      RSLO_CHECK_ARG(a.w[k] && a.occ[k] && a.pred[k - 1] && a.mpred[k - 1], "rslo_head_masks_fwd: null level pointer");
  }
  for (int k = 0; k < a.levels; ++k)
    hipLaunchKernelGGL(k_head_masks_fwd, dim3((unsigned)rslo_cdiv((int64_t)(a.H >> k) * (a.W >> k), 256), (unsigned)a.B),
                       dim3(256), 0, (hipStream_t)stream, a, k);
  RSLO_CHECK_LAUNCH("k_head_masks_fwd");
  return RSLO_OK;
}

extern "C" int rslo_head_masks_bwd(const RsloHeadMasksBwd *h_a, void *stream) {
  RSLO_CHECK_ARG(h_a && h_a->B >= 1 && h_a->B < 65536 && h_a->levels >= 1 && h_a->levels <= 4 && h_a->mask && h_a->d_tq,
                 "rslo_head_masks_bwd: bad arguments");
  HeadMaskBwdArgs a;
  a.mask = h_a->mask; a.g_mtq = h_a->g_mtq; a.d_tq = h_a->d_tq;
  a.B = h_a->B; a.H = h_a->H; a.W = h_a->W; a.levels = h_a->levels;
  for (int k = 0; k < 4; ++k) {
    a.occ[k] = h_a->occ[k];
    if (k < 3) {
      a.g_mpred[k] = h_a->g_mpred[k];
      a.d_pred[k] = h_a->d_pred[k];
    }
    if (k >= 1 && k < a.levels) RSLO_CHECK_ARG(a.occ[k] && a.d_pred[k - 1], "rslo_head_masks_bwd: null level pointer");
  }
  hipLaunchKernelGGL(k_head_masks_bwd, dim3((unsigned)rslo_cdiv((int64_t)a.H * a.W, 256), (unsigned)a.B, (unsigned)a.levels),
                     dim3(256), 0, (hipStream_t)stream, a);
  RSLO_CHECK_LAUNCH("k_head_masks_bwd");
  return RSLO_OK;
}

// ----------------------------------------------------------------------------------------- concatenation + upsampling
// The input of every deblock (odom_pred.py:219-221 of the reference: x = deblock(cat([x, skip], 1)), deblock =
// Upsample(s) -> Conv -> BN -> ReLU) is a channel concatenation that is then repeated s x s times: two launches and a
// (Ca + Cb) x H x W intermediate forward, an s x s window sum, two slices and their copies backward.  One launch each
// way: a thread takes one SOURCE cell, writes / sums its s x s window (window order = rows, then columns, from zero:
// the order of upsample_nearest2d_backward, so the sums have the same bits).
__global__ void k_cat_upsample_fwd(const float *__restrict__ a, const float *__restrict__ b, int Ca, int Cb, int H, int W,
                                   int s, int64_t n, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  int64_t r = i / W;
  const int y = (int)(r % H); r /= H;
  const int c = (int)(r % (Ca + Cb));
  const int64_t bb = r / (Ca + Cb);
  const float v = c < Ca ? a[((bb * Ca + c) * H + y) * W + x] : b[((bb * Cb + (c - Ca)) * H + y) * W + x];
  float *dst = out + (((bb * (Ca + Cb) + c) * H + y) * (int64_t)s) * W * s + (int64_t)x * s;
  for (int dy = 0; dy < s; ++dy)
    for (int dx = 0; dx < s; ++dx) dst[(int64_t)dy * W * s + dx] = v;
}

__global__ void k_cat_upsample_bwd(const float *__restrict__ g, int Ca, int Cb, int H, int W, int s, int64_t n,
                                   float *__restrict__ da, float *__restrict__ db) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  int64_t r = i / W;
  const int y = (int)(r % H); r /= H;
  const int c = (int)(r % (Ca + Cb));
  const int64_t bb = r / (Ca + Cb);
  const float *src = g + (((bb * (Ca + Cb) + c) * H + y) * (int64_t)s) * W * s + (int64_t)x * s;
  float acc = 0.f;
  for (int dy = 0; dy < s; ++dy)
    for (int dx = 0; dx < s; ++dx) acc = __fadd_rn(acc, src[(int64_t)dy * W * s + dx]);
  if (c < Ca) { if (da) da[((bb * Ca + c) * H + y) * W + x] = acc; }
  else if (db) db[((bb * Cb + (c - Ca)) * H + y) * W + x] = acc;
}

extern "C" int rslo_cat_upsample_fwd(const float *a, const float *b, int B, int Ca, int Cb, int H, int W, int scale, float *out,
                                     void *stream) {
  RSLO_CHECK_ARG(a && b && out && B >= 1 && Ca >= 1 && Cb >= 1 && H >= 1 && W >= 1 && scale >= 1 && scale <= 8,
                 "rslo_cat_upsample_fwd: bad arguments");
  const int64_t n = (int64_t)B * (Ca + Cb) * H * W;
  hipLaunchKernelGGL(k_cat_upsample_fwd, dim3((unsigned)rslo_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, Ca, Cb,
                     H, W, scale, n, out);
  RSLO_CHECK_LAUNCH("k_cat_upsample_fwd");
  return RSLO_OK;
}

extern "C" int rslo_cat_upsample_bwd(const float *grad, int B, int Ca, int Cb, int H, int W, int scale, float *da, float *db,
                                     void *stream) {
  RSLO_CHECK_ARG(grad && (da || db) && B >= 1 && Ca >= 1 && Cb >= 1 && H >= 1 && W >= 1 && scale >= 1 && scale <= 8,
                 "rslo_cat_upsample_bwd: bad arguments");
  const int64_t n = (int64_t)B * (Ca + Cb) * H * W;
  hipLaunchKernelGGL(k_cat_upsample_bwd, dim3((unsigned)rslo_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, grad, Ca, Cb,
                     H, W, scale, n, da, db);
  RSLO_CHECK_LAUNCH("k_cat_upsample_bwd");
  return RSLO_OK;
}

// ----------------------------------------------------------------------------------------- voted pose -> (t, q / (|q| + eps))
// odom [B,7] = (t, q) from the vote; the head returns t and the re-normalised quaternion q / (|q| + 1e-12)
// (odom_pred.py:279-288 of the reference).  As torch ops: 2 slices + norm + add + div forward and ~18 launches of their
// autograd backward on 28 numbers; here one launch each way (one thread per sample).
__global__ void k_pose_tail_fwd(const float *__restrict__ odom, int B, float *__restrict__ t, float *__restrict__ r) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *o = odom + (int64_t)b * 7;
#pragma unroll
  for (int k = 0; k < 3; ++k) t[b * 3 + k] = o[k];
  const float n = sqrtf(o[3] * o[3] + o[4] * o[4] + o[5] * o[5] + o[6] * o[6]);
  const float den = n + 1e-12f;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[b * 4 + k] = o[3 + k] / den;
}

// d_odom[:, :3] = g_t (zeros when NULL); d_odom[:, 3:] = g_r / den - q (q . g_r) / (den^2 |q|)   (the |q| term drops at
// |q| = 0, where torch.norm's gradient is defined as zero)
__global__ void k_pose_tail_bwd(const float *__restrict__ odom, const float *__restrict__ g_t,
                                const float *__restrict__ g_r, int B, float *__restrict__ d_odom) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *o = odom + (int64_t)b * 7;
  float *d = d_odom + (int64_t)b * 7;
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = g_t ? g_t[b * 3 + k] : 0.f;
  float q[4], g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { q[k] = o[3 + k]; g[k] = g_r ? g_r[b * 4 + k] : 0.f; }
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float den = n + 1e-12f;
  const float s = q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3];
  const float c = n > 0.f ? s / (den * den * n) : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) d[3 + k] = g[k] / den - q[k] * c;
}

extern "C" int rslo_pose_tail_fwd(const float *odom, int B, float *t, float *r, void *stream) {
  RSLO_CHECK_ARG(odom && t && r && B >= 1, "rslo_pose_tail_fwd: bad arguments");
  hipLaunchKernelGGL(k_pose_tail_fwd, dim3((unsigned)rslo_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, odom, B, t, r);
  RSLO_CHECK_LAUNCH("k_pose_tail_fwd");
  return RSLO_OK;
}

extern "C" int rslo_pose_tail_bwd(const float *odom, const float *g_t, const float *g_r, int B, float *d_odom, void *stream) {
  RSLO_CHECK_ARG(odom && d_odom && B >= 1, "rslo_pose_tail_bwd: bad arguments");
  hipLaunchKernelGGL(k_pose_tail_bwd, dim3((unsigned)rslo_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, odom, g_t, g_r, B,
                     d_odom);
  RSLO_CHECK_LAUNCH("k_pose_tail_bwd");
  return RSLO_OK;
}
