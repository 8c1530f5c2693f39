// Per-frame (segmented) BatchNorm1d + LeakyReLU for the covariance branch of the sparse encoder
// (reference: the raw nn.BatchNorm1d modules at rslo/models/middle.py:181-198, followed by nn.LeakyReLU).
//
// The reference runs the encoder one frame at a time, so batch statistics are per frame.  Here all frames are
// one batched tensor whose rows are grouped by frame (seg_off[S+1]); each segment gets its own mean / biased
// variance, and the running estimates are updated segment after segment in frame order exactly like S
// consecutive module calls (momentum update with the unbiased variance).  Three launches forward and three
// backward per layer instead of ~16 torch kernels per frame.  HBM-bound elementwise/reduction work; partial
// sums in double, fixed reduction order (deterministic).
#include "rslo_common.h"

#define SB_THREADS 256
#define SB_ROWS 256   /* rows per chunk */

__device__ __forceinline__ int sb_pad(int C) { return C <= 16 ? 16 : (C <= 32 ? 32 : 64); }

// partial[(seg * nchunk + chunk) * 2C + {c, C + c}] = sum x, sum x^2   (forward)
//                                                    = sum g, sum g*xhat (backward; g = gy * act'(y))
template <bool BWD>
__global__ __launch_bounds__(SB_THREADS) void k_segbn_partial(const float *__restrict__ x, const float *__restrict__ y,
                                                              const float *__restrict__ gy, int C,
                                                              const int32_t *__restrict__ seg_off, int nchunk,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, float slope,
                                                              double *__restrict__ partial) {
  __shared__ double red[2][SB_THREADS];
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const int CP = sb_pad(C), c = threadIdx.x % CP, part = threadIdx.x / CP, nparts = SB_THREADS / CP;
  const int64_t r0 = (int64_t)seg_off[seg] + (int64_t)chunk * SB_ROWS;
  const int64_t rend = seg_off[seg + 1];
  const int64_t r1 = (r0 + SB_ROWS < rend) ? r0 + SB_ROWS : rend;
  double s0 = 0.0, s1 = 0.0;
  if (c < C && r0 < rend) {
    const float mu = BWD ? mean[seg * C + c] : 0.f, is = BWD ? invstd[seg * C + c] : 0.f;
    for (int64_t r = r0 + part; r < r1; r += nparts) {
      const float xv = x[r * C + c];
      if (BWD) {
        float g = gy[r * C + c];
        if (slope != 1.0f && !(y[r * C + c] > 0.f)) g *= slope;
        s0 += g;
        s1 += (double)g * (double)((xv - mu) * is);
      } else {
        s0 += xv;
        s1 += (double)xv * (double)xv;
      }
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (part == 0 && c < C) {
    double a = 0.0, b = 0.0;
    for (int p = 0; p < nparts; ++p) {
      a += red[0][p * CP + c];
      b += red[1][p * CP + c];
    }
    double *o = partial + ((int64_t)seg * nchunk + chunk) * 2 * C;
    o[c] = a;
    o[C + c] = b;
  }
}

// forward finish: ONE block; SBF_SLOTS groups of SB_THREADS threads take SBF_SLOTS consecutive segments at a time (each
// group sums its segment's chunk partials exactly as a lone group would: same parts, same order, same bits), then the
// first group applies the running-estimate updates of those segments one after the other in frame order.  (A single
// group walking the 8 frames of a step one by one took 24-27 us per launch, 10 launches per step.)
#define SBF_SLOTS 4
__global__ __launch_bounds__(SB_THREADS * SBF_SLOTS) void k_segbn_fwd_finish(const double *__restrict__ partial, int C, int S,
                                                                 const int32_t *__restrict__ seg_off, int nchunk,
                                                                 float eps, float momentum,
                                                                 float *__restrict__ running_mean,
                                                                 float *__restrict__ running_var,
                                                                 float *__restrict__ mean, float *__restrict__ invstd) {
  __shared__ double red[2][SB_THREADS * SBF_SLOTS];
  __shared__ double st_mu[SBF_SLOTS][64], st_unb[SBF_SLOTS][64];
  const int slot = threadIdx.x / SB_THREADS, t = threadIdx.x % SB_THREADS;
  const int CP = sb_pad(C), c = t % CP, part = t / CP, nparts = SB_THREADS / CP;
  for (int s0 = 0; s0 < S; s0 += SBF_SLOTS) {
    const int seg = s0 + slot;
    const bool live = seg < S;
    const int64_t n = live ? (int64_t)seg_off[seg + 1] - seg_off[seg] : 0;
    const int used = (int)((n + SB_ROWS - 1) / SB_ROWS);
    double a = 0.0, b = 0.0;
    if (live && c < C)
      for (int k = part; k < used; k += nparts) {
        const double *o = partial + ((int64_t)seg * nchunk + k) * 2 * C;
        a += o[c];
        b += o[C + c];
      }
    __syncthreads();
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (live && part == 0 && c < C && n > 0) {
      double sa = 0.0, sb = 0.0;
      for (int p = 0; p < nparts; ++p) {
        sa += red[0][slot * SB_THREADS + p * CP + c];
        sb += red[1][slot * SB_THREADS + p * CP + c];
      }
      const double mu = sa / (double)n;
      double var = sb / (double)n - mu * mu;
      if (var < 0.0) var = 0.0;
      mean[seg * C + c] = (float)mu;
      invstd[seg * C + c] = (float)(1.0 / sqrt(var + (double)eps));
      st_mu[slot][c] = mu;
      st_unb[slot][c] = n > 1 ? var * (double)n / (double)(n - 1) : var;
    }
    __syncthreads();
    if (running_mean && slot == 0 && part == 0 && c < C)
      for (int j = 0; j < SBF_SLOTS && s0 + j < S; ++j) {
        if (seg_off[s0 + j + 1] - seg_off[s0 + j] <= 0) continue;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * st_mu[j][c]);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * st_unb[j][c]);
      }
  }
}

__global__ __launch_bounds__(SB_THREADS) void k_segbn_fwd_apply(const float *__restrict__ x, int C,
                                                                const int32_t *__restrict__ seg_off,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float slope,
                                                                float *__restrict__ y) {
  const int seg = blockIdx.y;
  const int64_t r0 = (int64_t)seg_off[seg] + (int64_t)blockIdx.x * SB_ROWS;
  const int64_t rend = seg_off[seg + 1];
  const int64_t e0 = r0 * C, e1 = ((r0 + SB_ROWS < rend) ? r0 + SB_ROWS : rend) * C;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += SB_THREADS) {
    const int c = (int)(e % C);
    float v = (x[e] - mean[seg * C + c]) * invstd[seg * C + c];
    v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
    y[e] = v > 0.f ? v : v * slope;
  }
}

// backward finish: per-segment sums -> sg[S,C], sgx[S,C]; dgamma = sum_s sgx, dbeta = sum_s sg
__global__ __launch_bounds__(SB_THREADS * SBF_SLOTS) void k_segbn_bwd_finish(const double *__restrict__ partial, int C, int S,
                                                                 const int32_t *__restrict__ seg_off, int nchunk,
                                                                 float *__restrict__ sg, float *__restrict__ sgx,
                                                                 float *__restrict__ dgamma, float *__restrict__ dbeta) {
  __shared__ double red[2][SB_THREADS * SBF_SLOTS];
  __shared__ double st_a[SBF_SLOTS][64], st_b[SBF_SLOTS][64];
  const int slot = threadIdx.x / SB_THREADS, t = threadIdx.x % SB_THREADS;
  const int CP = sb_pad(C), c = t % CP, part = t / CP, nparts = SB_THREADS / CP;
  double tg = 0.0, tb = 0.0;      // over the segments in frame order (first group only)
  for (int s0 = 0; s0 < S; s0 += SBF_SLOTS) {
    const int seg = s0 + slot;
    const bool live = seg < S;
    const int64_t n = live ? (int64_t)seg_off[seg + 1] - seg_off[seg] : 0;
    const int used = (int)((n + SB_ROWS - 1) / SB_ROWS);
    double a = 0.0, b = 0.0;
    if (live && c < C)
      for (int k = part; k < used; k += nparts) {
        const double *o = partial + ((int64_t)seg * nchunk + k) * 2 * C;
        a += o[c];
        b += o[C + c];
      }
    __syncthreads();
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (live && part == 0 && c < C) {
      double sa = 0.0, sb = 0.0;
      for (int p = 0; p < nparts; ++p) {
        sa += red[0][slot * SB_THREADS + p * CP + c];
        sb += red[1][slot * SB_THREADS + p * CP + c];
      }
      sg[seg * C + c] = (float)(n > 0 ? sa / (double)n : 0.0);     // mean of g
      sgx[seg * C + c] = (float)(n > 0 ? sb / (double)n : 0.0);    // mean of g * xhat
      st_a[slot][c] = sa;
      st_b[slot][c] = sb;
    }
    __syncthreads();
    if (slot == 0 && part == 0 && c < C)
      for (int j = 0; j < SBF_SLOTS && s0 + j < S; ++j) {
        tb += st_a[j][c];
        tg += st_b[j][c];
      }
  }
  if (slot == 0 && part == 0 && c < C) {
    if (dgamma) dgamma[c] = (float)tg;
    if (dbeta) dbeta[c] = (float)tb;
  }
}

__global__ __launch_bounds__(SB_THREADS) void k_segbn_bwd_apply(const float *__restrict__ x, const float *__restrict__ y,
                                                                const float *__restrict__ gy, int C,
                                                                const int32_t *__restrict__ seg_off,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ sg,
                                                                const float *__restrict__ sgx, float slope,
                                                                float *__restrict__ gx) {
  const int seg = blockIdx.y;
  const int64_t r0 = (int64_t)seg_off[seg] + (int64_t)blockIdx.x * SB_ROWS;
  const int64_t rend = seg_off[seg + 1];
  const int64_t e0 = r0 * C, e1 = ((r0 + SB_ROWS < rend) ? r0 + SB_ROWS : rend) * C;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += SB_THREADS) {
    const int c = (int)(e % C);
    float g = gy[e];
    if (slope != 1.0f && !(y[e] > 0.f)) g *= slope;
    const float is = invstd[seg * C + c];
    const float xh = (x[e] - mean[seg * C + c]) * is;
    gx[e] = (gamma ? gamma[c] : 1.f) * is * (g - sg[seg * C + c] - xh * sgx[seg * C + c]);
  }
}

static int sb_chunks(int64_t max_seg_len) { return (int)rslo_cdiv(max_seg_len > 0 ? max_seg_len : 1, SB_ROWS); }

extern "C" size_t rslo_segbn_ws_bytes(int S, int64_t max_seg_len, int C) {
  return (size_t)(S > 0 ? S : 1) * sb_chunks(max_seg_len) * 2 * C * sizeof(double);
}

extern "C" int rslo_segbn_fwd(const float *x, int C, const int32_t *seg_off, int S, int64_t max_seg_len,
                              const float *gamma, const float *beta, float *running_mean, float *running_var,
                              float momentum, float eps, float act_slope, void *ws, size_t ws_bytes, float *y,
                              float *save_mean, float *save_invstd, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(C >= 1 && C <= 64, "segbn: C must be in 1..64");
  if (S == 0 || max_seg_len == 0) return RSLO_OK;
  if (ws_bytes < rslo_segbn_ws_bytes(S, max_seg_len, C)) {
    rslo_set_error("segbn_fwd: workspace too small");
    return RSLO_EWS;
  }
  const int nch = sb_chunks(max_seg_len);
  dim3 grid((unsigned)nch, (unsigned)S);
  hipLaunchKernelGGL(k_segbn_partial<false>, grid, dim3(SB_THREADS), 0, st, x, nullptr, nullptr, C, seg_off, nch,
                     nullptr, nullptr, 1.0f, (double *)ws);
  hipLaunchKernelGGL(k_segbn_fwd_finish, dim3(1), dim3(SB_THREADS * SBF_SLOTS), 0, st, (const double *)ws, C, S, seg_off, nch,
                     eps, momentum, running_mean, running_var, save_mean, save_invstd);
  hipLaunchKernelGGL(k_segbn_fwd_apply, grid, dim3(SB_THREADS), 0, st, x, C, seg_off, save_mean, save_invstd, gamma,
                     beta, act_slope, y);
  RSLO_CHECK_LAUNCH("segbn_fwd");
  return RSLO_OK;
}

extern "C" int rslo_segbn_bwd(const float *x, const float *y, const float *gy, int C, const int32_t *seg_off, int S,
                              int64_t max_seg_len, const float *gamma, const float *save_mean,
                              const float *save_invstd, float act_slope, void *ws, size_t ws_bytes, float *gx,
                              float *dgamma, float *dbeta, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(C >= 1 && C <= 64, "segbn: C must be in 1..64");
  if (S == 0 || max_seg_len == 0) return RSLO_OK;
  const size_t need = rslo_segbn_ws_bytes(S, max_seg_len, C) + (size_t)2 * S * C * sizeof(float);
  if (ws_bytes < need) {
    rslo_set_error("segbn_bwd: workspace too small");
    return RSLO_EWS;
  }
  const int nch = sb_chunks(max_seg_len);
  double *partial = (double *)ws;
  float *sg = (float *)((char *)ws + rslo_segbn_ws_bytes(S, max_seg_len, C));
  float *sgx = sg + (size_t)S * C;
  dim3 grid((unsigned)nch, (unsigned)S);
  hipLaunchKernelGGL(k_segbn_partial<true>, grid, dim3(SB_THREADS), 0, st, x, y, gy, C, seg_off, nch, save_mean,
                     save_invstd, act_slope, partial);
  hipLaunchKernelGGL(k_segbn_bwd_finish, dim3(1), dim3(SB_THREADS * SBF_SLOTS), 0, st, (const double *)partial, C, S, seg_off,
                     nch, sg, sgx, dgamma, dbeta);
  hipLaunchKernelGGL(k_segbn_bwd_apply, grid, dim3(SB_THREADS), 0, st, x, y, gy, C, seg_off, save_mean, save_invstd,
                     gamma, sg, sgx, act_slope, gx);
  RSLO_CHECK_LAUNCH("segbn_bwd");
  return RSLO_OK;
}


// ---- eval mode (evaluate.py:363-408 runs the network in eval(): rslo/models/middle.py:181-213's nn.BatchNorm1d layers then
// normalise with their running statistics) -- y = act((x - mean) / sqrt(var + eps) * gamma + beta), one launch instead of the
// library's statistics-to-invstd kernel + transform kernel + a separate activation kernel.  n_live: optional device count of
// the live rows of a capacity-laid-out tensor (rows past it are padding: left unwritten).
__global__ __launch_bounds__(256) void k_bn1d_eval_act(const float *__restrict__ x, int64_t n, int C, const float *__restrict__ mean,
                                                       const float *__restrict__ var, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float eps, float slope,
                                                       const int32_t *__restrict__ n_live, float *__restrict__ y) {
  const int64_t rows = n_live ? (int64_t)*n_live : n;
  const int64_t total = (rows < n ? rows : n) * C;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = i + e;
      if (j < total) {
        const int c = (int)(j % C);
        const float inv = 1.0f / sqrtf(var[c] + eps);
        float v = (x[j] - mean[c]) * inv;
        v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
        y[j] = v > 0.f ? v : v * slope;
      }
    }
  }
}

extern "C" int rslo_bn1d_eval_act(const float *x, int64_t n, int C, const float *running_mean, const float *running_var,
                                  const float *gamma, const float *beta, float eps, float act_slope, const int32_t *n_live_dev,
                                  float *y, void *stream) {
  RSLO_CHECK_ARG(x && y && running_mean && running_var && n >= 0 && C >= 1, "rslo_bn1d_eval_act: bad arguments");
  if (n == 0) return RSLO_OK;
  int64_t blocks = rslo_cdiv(n * C, 1024);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_bn1d_eval_act, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, C, running_mean,
                     running_var, gamma, beta, eps, act_slope, n_live_dev, y);
  RSLO_CHECK_LAUNCH("k_bn1d_eval_act");
  return RSLO_OK;
}
