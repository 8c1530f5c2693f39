// Training-mode (Sync)BatchNorm2d of the dense BEV head fused with its activation and residual add, NCHW fp32
// (reference: apex.parallel.SyncBatchNorm used through rslo/layers/SparseConv.py:96-113 in
//  rslo/models/custom_resnet_spc.py:224-298 and the conv-BN-ReLU stacks of rslo/models/odom_pred.py).
//
// One code path for 1..N ranks: [stats kernel] -> (all-reduce of 2C+1 doubles when the process group has more than one
// rank) -> [apply kernel: normalise + affine + residual + (Leaky)ReLU, running statistics, saved mean / invstd];
// backward: [reduce kernel: sum g, sum g x^ with g = dy * act'(y)] -> (all-reduce of 2C doubles) -> [apply kernel: dx,
// d_residual].  Two launches per direction instead of BN + activation + add (three kernels and three autograd nodes),
// and across ranks ONE small collective per direction instead of torch SyncBatchNorm's gather / reduce sequences.
// Sums are accumulated in double; partial sums are added in block order by the last block of a channel: deterministic.
// HBM-bound elementwise work: forward reads x twice (+ residual) and writes y; backward reads dy, y, x twice, writes dx.
#include <stdlib.h>

#include "peer_comm.h"

#define BN_THREADS 256

__device__ __forceinline__ void bn_block_sum2(double a, double b, double *out2) {
  __shared__ double red[2][BN_THREADS / 64];
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_down(a, o, 64);
    b += __shfl_down(b, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int w = 0; w < BN_THREADS / 64; ++w) {
      s0 += red[0][w];
      s1 += red[1][w];
    }
    out2[0] = s0;
    out2[1] = s1;
  }
  __syncthreads();
}

// grid (SPLIT, C).  part [C][SPLIT][2]; done [C] zero on entry / exit; stats [2C+1] = sum[c], sumsq[c], count.
__global__ __launch_bounds__(BN_THREADS) void k_bn2d_stats(const float *__restrict__ x, int N, int C, int HW,
                                                           int64_t per_blk, double *__restrict__ part,
                                                           int *__restrict__ done, double *__restrict__ stats) {
  const int c = blockIdx.y, sp = blockIdx.x, S = gridDim.x;
  const int64_t total = (int64_t)N * HW;
  const int64_t e0 = (int64_t)sp * per_blk, e1 = (e0 + per_blk < total) ? e0 + per_blk : total;
  double s = 0.0, q = 0.0;
  for (int64_t n = e0 / HW; n * HW < e1; ++n) {        // the slice [e0, e1) row by row: no per-element division
    const int64_t lo = (e0 > n * HW ? e0 - n * HW : 0), hi = (e1 < (n + 1) * HW ? e1 - n * HW : HW);
    const float *row = x + (n * C + c) * (int64_t)HW;
    if (((HW | lo) & 3) == 0) {
      for (int64_t k = lo + 4 * threadIdx.x; k + 3 < hi; k += 4 * BN_THREADS) {
        const float4 v = *reinterpret_cast<const float4 *>(row + k);
        s += ((double)v.x + v.y) + ((double)v.z + v.w);
        q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
      }
      const int64_t tail = lo + ((hi - lo) & ~(int64_t)3);
      for (int64_t k = tail + threadIdx.x; k < hi; k += BN_THREADS) {
        s += row[k];
        q += (double)row[k] * row[k];
      }
    } else {
      for (int64_t k = lo + threadIdx.x; k < hi; k += BN_THREADS) {
        s += row[k];
        q += (double)row[k] * row[k];
      }
    }
  }
  __shared__ double res[2];
  bn_block_sum2(s, q, res);
  if (threadIdx.x == 0) {
    part[((int64_t)c * S + sp) * 2 + 0] = res[0];
    part[((int64_t)c * S + sp) * 2 + 1] = res[1];
  }
}

// Second stage of both reductions: one thread per channel adds its S slice partials in slice order.  A separate launch
// instead of a last-block finish: the device-scope fence a last-block scheme needs writes back the XCD's L2, which on
// this part costs ~10 us per launch right after a convolution has filled it with dirty lines (measured: 12-24 us per
// reduction with the fence).
__global__ void k_bn2d_finish(const double *__restrict__ part, int S, int C, double count, double *__restrict__ out,
                              float *__restrict__ fa, float *__restrict__ fb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < S; ++k) {
    a += part[((int64_t)c * S + k) * 2 + 0];
    b += part[((int64_t)c * S + k) * 2 + 1];
  }
  out[c] = a;
  out[C + c] = b;
  if (count >= 0.0 && c == 0) out[2 * C] = count;
  if (fa) fa[c] = (float)a;
  if (fb) fb[c] = (float)b;
}

// y = act(gamma * (x - mean) * invstd + beta (+ res)); one thread per 4 consecutive hw of one (n, c) row when HW % 4 == 0
__global__ __launch_bounds__(BN_THREADS) void k_bn2d_apply(const float *__restrict__ x, const float *__restrict__ res,
                                                           const double *__restrict__ stats,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           int N, int C, int HW, float eps, float momentum, float slope,
                                                           float *__restrict__ run_mean, float *__restrict__ run_var,
                                                           float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                           float *__restrict__ y, const double *__restrict__ part,
                                                           int S) {
  const int row = blockIdx.y;                 // n * C + c
  const int c = row % C;
  double cnt, sum, sumsq;
  if (part) {                                 // single rank: add the slice partials here (no finish launch)
    cnt = (double)N * HW;
    sum = sumsq = 0.0;
    for (int k = 0; k < S; ++k) {
      sum += part[((int64_t)c * S + k) * 2 + 0];
      sumsq += part[((int64_t)c * S + k) * 2 + 1];
    }
  } else {
    cnt = stats[2 * C];
    sum = stats[c];
    sumsq = stats[C + c];
  }
  const double m = sum / cnt;
  double var = sumsq / cnt - m * m;
  var = var > 0.0 ? var : 0.0;
  const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float a1 = g * invstd, a0 = b - mean * a1;
  if (row < C && blockIdx.x == 0 && threadIdx.x == 0) {   // n == 0: one thread per channel keeps the statistics
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (run_mean) {
      const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
  const int64_t base = (int64_t)row * HW;
  const int hw = (blockIdx.x * BN_THREADS + threadIdx.x) * 4;
  if (hw >= HW) return;
  if ((HW & 3) == 0) {
    float4 v = *reinterpret_cast<const float4 *>(x + base + hw);
    float4 r = res ? *reinterpret_cast<const float4 *>(res + base + hw) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    o.x = v.x * a1 + a0 + r.x;
    o.y = v.y * a1 + a0 + r.y;
    o.z = v.z * a1 + a0 + r.z;
    o.w = v.w * a1 + a0 + r.w;
    o.x = o.x > 0.f ? o.x : o.x * slope;
    o.y = o.y > 0.f ? o.y : o.y * slope;
    o.z = o.z > 0.f ? o.z : o.z * slope;
    o.w = o.w > 0.f ? o.w : o.w * slope;
    *reinterpret_cast<float4 *>(y + base + hw) = o;
  } else {
    for (int k = hw; k < hw + 4 && k < HW; ++k) {
      float o = x[base + k] * a1 + a0 + (res ? res[base + k] : 0.f);
      y[base + k] = o > 0.f ? o : o * slope;
    }
  }
}

// sums of g = dy * act'(y) and g * x^ per channel.  red [2C] (for the cross-rank sum), dgamma / dbeta = the LOCAL sums.
__global__ __launch_bounds__(BN_THREADS) void k_bn2d_bwd_reduce(const float *__restrict__ dy, const float *__restrict__ y,
                                                                const float *__restrict__ x,
                                                                const float *__restrict__ save_mean,
                                                                const float *__restrict__ save_invstd, int N, int C,
                                                                int HW, int64_t per_blk, float slope, int has_act,
                                                                double *__restrict__ part, int *__restrict__ done,
                                                                double *__restrict__ red, float *__restrict__ dgamma,
                                                                float *__restrict__ dbeta) {
  const int c = blockIdx.y, sp = blockIdx.x, S = gridDim.x;
  const int64_t total = (int64_t)N * HW;
  const int64_t e0 = (int64_t)sp * per_blk, e1 = (e0 + per_blk < total) ? e0 + per_blk : total;
  const float mean = save_mean[c], invstd = save_invstd[c];
  double s = 0.0, q = 0.0;
  for (int64_t n = e0 / HW; n * HW < e1; ++n) {
    const int64_t lo = (e0 > n * HW ? e0 - n * HW : 0), hi = (e1 < (n + 1) * HW ? e1 - n * HW : HW);
    const int64_t base = (n * C + c) * (int64_t)HW;
    if (((HW | lo) & 3) == 0) {
      for (int64_t k = lo + 4 * threadIdx.x; k + 3 < hi; k += 4 * BN_THREADS) {
        const float4 d = *reinterpret_cast<const float4 *>(dy + base + k);
        const float4 v = *reinterpret_cast<const float4 *>(x + base + k);
        float g[4] = {d.x, d.y, d.z, d.w};
        if (has_act) {
          const float4 o = *reinterpret_cast<const float4 *>(y + base + k);
          g[0] = o.x > 0.f ? g[0] : g[0] * slope;
          g[1] = o.y > 0.f ? g[1] : g[1] * slope;
          g[2] = o.z > 0.f ? g[2] : g[2] * slope;
          g[3] = o.w > 0.f ? g[3] : g[3] * slope;
        }
        const float xh[4] = {(v.x - mean) * invstd, (v.y - mean) * invstd, (v.z - mean) * invstd, (v.w - mean) * invstd};
        s += ((double)g[0] + g[1]) + ((double)g[2] + g[3]);
        q += ((double)g[0] * xh[0] + (double)g[1] * xh[1]) + ((double)g[2] * xh[2] + (double)g[3] * xh[3]);
      }
      const int64_t tail = lo + ((hi - lo) & ~(int64_t)3);
      for (int64_t k = tail + threadIdx.x; k < hi; k += BN_THREADS) {
        float g = dy[base + k];
        if (has_act) g = y[base + k] > 0.f ? g : g * slope;
        s += g;
        q += (double)g * (double)((x[base + k] - mean) * invstd);
      }
    } else {
      for (int64_t k = lo + threadIdx.x; k < hi; k += BN_THREADS) {
        float g = dy[base + k];
        if (has_act) g = y[base + k] > 0.f ? g : g * slope;
        s += g;
        q += (double)g * (double)((x[base + k] - mean) * invstd);
      }
    }
  }
  __shared__ double res[2];
  bn_block_sum2(s, q, res);
  if (threadIdx.x == 0) {
    part[((int64_t)c * S + sp) * 2 + 0] = res[0];
    part[((int64_t)c * S + sp) * 2 + 1] = res[1];
  }
}

// dx = gamma invstd (g - sum_g / cnt - x^ sum_gx / cnt);  dres = g
__global__ __launch_bounds__(BN_THREADS) void k_bn2d_bwd_apply(const float *__restrict__ dy, const float *__restrict__ y,
                                                               const float *__restrict__ x,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ save_mean,
                                                               const float *__restrict__ save_invstd,
                                                               const double *__restrict__ red, double cnt,
                                                               const double *__restrict__ cnt_dev, int C, int HW,
                                                               float slope, int has_act, float *__restrict__ dx,
                                                               float *__restrict__ dres, const double *__restrict__ part,
                                                               int S, float *__restrict__ dgamma,
                                                               float *__restrict__ dbeta) {
  const int row = blockIdx.y, c = row % C;
  const float mean = save_mean[c], invstd = save_invstd[c];
  const float k0 = (gamma ? gamma[c] : 1.f) * invstd;
  double ra, rb;
  if (part) {                                 // single rank: add the slice partials here (no finish launch)
    ra = rb = 0.0;
    for (int k = 0; k < S; ++k) {
      ra += part[((int64_t)c * S + k) * 2 + 0];
      rb += part[((int64_t)c * S + k) * 2 + 1];
    }
    if (row < C && blockIdx.x == 0 && threadIdx.x == 0) {
      if (dbeta) dbeta[c] = (float)ra;
      if (dgamma) dgamma[c] = (float)rb;
    }
  } else {
    ra = red[c];
    rb = red[C + c];
  }
  if (cnt_dev) cnt = *cnt_dev;               // element count over all ranks, as all-reduced in the forward pass
  const float mg = (float)(ra / cnt), mgx = (float)(rb / cnt);
  const int64_t base = (int64_t)row * HW;
  const int hw0 = (blockIdx.x * BN_THREADS + threadIdx.x) * 4;
  if (hw0 >= HW) return;
  if ((HW & 3) == 0) {
    const float4 d = *reinterpret_cast<const float4 *>(dy + base + hw0);
    const float4 v = *reinterpret_cast<const float4 *>(x + base + hw0);
    float g[4] = {d.x, d.y, d.z, d.w};
    if (has_act) {
      const float4 o = *reinterpret_cast<const float4 *>(y + base + hw0);
      g[0] = o.x > 0.f ? g[0] : g[0] * slope;
      g[1] = o.y > 0.f ? g[1] : g[1] * slope;
      g[2] = o.z > 0.f ? g[2] : g[2] * slope;
      g[3] = o.w > 0.f ? g[3] : g[3] * slope;
    }
    float4 r;
    r.x = k0 * (g[0] - mg - (v.x - mean) * invstd * mgx);
    r.y = k0 * (g[1] - mg - (v.y - mean) * invstd * mgx);
    r.z = k0 * (g[2] - mg - (v.z - mean) * invstd * mgx);
    r.w = k0 * (g[3] - mg - (v.w - mean) * invstd * mgx);
    *reinterpret_cast<float4 *>(dx + base + hw0) = r;
    if (dres) *reinterpret_cast<float4 *>(dres + base + hw0) = make_float4(g[0], g[1], g[2], g[3]);
    return;
  }
  for (int k = hw0; k < hw0 + 4 && k < HW; ++k) {
    float g = dy[base + k];
    if (has_act) g = y[base + k] > 0.f ? g : g * slope;
    const float xh = (x[base + k] - mean) * invstd;
    dx[base + k] = k0 * (g - mg - xh * mgx);
    if (dres) dres[base + k] = g;
  }
}

// ---- single-launch forms for maps that fit one workgroup per channel (N * HW <= BN_SMALL_MAX elements: the 24x44 and 12x22 maps) ----------------
// One block owns a channel: pass 1 accumulates the sums (double, fixed order), pass 2 re-reads the (cache-resident)
// rows and applies.  One launch per direction instead of two: on this path the step is bound by kernel-launch cost on
// the host (~7 us per launch), not by the 5 us these kernels run.
#define BN_SMALL_MAX 17408      /* 17 x 1024: the 48x88 maps at bs 4 (16896 values per channel) included */

template <int T>
__device__ __forceinline__ void bn_block_sum2_t(double a, double b, double *out2) {
  __shared__ double red[2][T / 64];
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_down(a, o, 64);
    b += __shfl_down(b, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int w = 0; w < T / 64; ++w) {
      s0 += red[0][w];
      s1 += red[1][w];
    }
    out2[0] = s0;
    out2[1] = s1;
  }
  __syncthreads();
}

template <int T>
__global__ __launch_bounds__(T) void k_bn2d_fwd_small(const float *__restrict__ x, const float *__restrict__ res,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      int N, int C, int HW, float eps, float momentum, float slope,
                                                      float *__restrict__ run_mean, float *__restrict__ run_var,
                                                      float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                      float *__restrict__ y) {
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int n = 0; n < N; ++n) {
    const float *row = x + ((int64_t)n * C + c) * HW;
    for (int k = threadIdx.x; k < HW; k += T) {
      const float v = row[k];
      s += v;
      q += (double)v * v;
    }
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q, tot);
  const double cnt = (double)N * HW;
  const double m = tot[0] / cnt;
  double var = tot[1] / cnt - m * m;
  var = var > 0.0 ? var : 0.0;
  const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (run_mean) {
      const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float a1 = g * invstd, a0 = b - mean * a1;
  for (int n = 0; n < N; ++n) {
    const int64_t base = ((int64_t)n * C + c) * HW;
    for (int k = threadIdx.x; k < HW; k += T) {
      float o = x[base + k] * a1 + a0 + (res ? res[base + k] : 0.f);
      y[base + k] = o > 0.f ? o : o * slope;
    }
  }
}

template <int T>
__global__ __launch_bounds__(T) void k_bn2d_bwd_small(const float *__restrict__ dy, const float *__restrict__ y,
                                                      const float *__restrict__ x, const float *__restrict__ gamma,
                                                      const float *__restrict__ save_mean,
                                                      const float *__restrict__ save_invstd, int N, int C, int HW,
                                                      float slope, int has_act, float *__restrict__ dx,
                                                      float *__restrict__ dres, float *__restrict__ dgamma,
                                                      float *__restrict__ dbeta) {
  const int c = blockIdx.x;
  const float mean = save_mean[c], invstd = save_invstd[c];
  double s = 0.0, q = 0.0;
  for (int n = 0; n < N; ++n) {
    const int64_t base = ((int64_t)n * C + c) * HW;
    for (int k = threadIdx.x; k < HW; k += T) {
      float g = dy[base + k];
      if (has_act) g = y[base + k] > 0.f ? g : g * slope;
      s += g;
      q += (double)g * (double)((x[base + k] - mean) * invstd);
    }
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q, tot);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] = (float)tot[0];
    if (dgamma) dgamma[c] = (float)tot[1];
  }
  const double cnt = (double)N * HW;
  const float k0 = (gamma ? gamma[c] : 1.f) * invstd;
  const float mg = (float)(tot[0] / cnt), mgx = (float)(tot[1] / cnt);
  for (int n = 0; n < N; ++n) {
    const int64_t base = ((int64_t)n * C + c) * HW;
    for (int k = threadIdx.x; k < HW; k += T) {
      float g = dy[base + k];
      if (has_act) g = y[base + k] > 0.f ? g : g * slope;
      const float xh = (x[base + k] - mean) * invstd;
      dx[base + k] = k0 * (g - mg - xh * mgx);
      if (dres) dres[base + k] = g;
    }
  }
}

// Register-cached forms for channels of at most 8 T elements (the 24x44 and 12x22 maps at bs 4: 4.1 elements per thread):
// every element is loaded ONCE, all of a thread's loads are in flight together, and the second pass works from
// registers -- the two-pass loops above are two serial rounds of dependent-latency loads (12.5 us per launch for 17 KB
// per channel).  Element e = tid + i T of the channel's N * HW values (n = e / HW).
#define BN_RC 8
#define BN_RC_BIG 17     /* T = 1024: channels of up to 17408 values (48x88 at bs 4) in one launch instead of reduce + apply */
template <int T, int E = BN_RC>
__global__ __launch_bounds__(T) void k_bn2d_fwd_small_rc(const float *__restrict__ x, const float *__restrict__ res,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         int N, int C, int HW, float eps, float momentum, float slope,
                                                         float *__restrict__ run_mean, float *__restrict__ run_var,
                                                         float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                         float *__restrict__ y) {
  const int c = blockIdx.x, total = N * HW;
  float v[E], r[E];
  int64_t off[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int e = threadIdx.x + i * T;
    const int n = e / HW, k = e - n * HW;
    off[i] = e < total ? ((int64_t)n * C + c) * HW + k : -1;
    v[i] = off[i] >= 0 ? x[off[i]] : 0.f;
    r[i] = (res && off[i] >= 0) ? res[off[i]] : 0.f;
  }
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    s += v[i];
    q += (double)v[i] * v[i];
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q, tot);
  const double cnt = (double)N * HW;
  const double m = tot[0] / cnt;
  double var = tot[1] / cnt - m * m;
  var = var > 0.0 ? var : 0.0;
  const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (run_mean) {
      const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float a1 = g * invstd, a0 = b - mean * a1;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    if (off[i] < 0) continue;
    const float o = v[i] * a1 + a0 + r[i];
    y[off[i]] = o > 0.f ? o : o * slope;
  }
}

template <int T, int E = BN_RC>
__global__ __launch_bounds__(T) void k_bn2d_bwd_small_rc(const float *__restrict__ dy, const float *__restrict__ y,
                                                         const float *__restrict__ x, const float *__restrict__ gamma,
                                                         const float *__restrict__ save_mean,
                                                         const float *__restrict__ save_invstd, int N, int C, int HW,
                                                         float slope, int has_act, float *__restrict__ dx,
                                                         float *__restrict__ dres, float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta) {
  const int c = blockIdx.x, total = N * HW;
  const float mean = save_mean[c], invstd = save_invstd[c];
  float g[E], xh[E];
  int64_t off[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int e = threadIdx.x + i * T;
    const int n = e / HW, k = e - n * HW;
    off[i] = e < total ? ((int64_t)n * C + c) * HW + k : -1;
    const float gv = off[i] >= 0 ? dy[off[i]] : 0.f;
    const float yv = (has_act && off[i] >= 0) ? y[off[i]] : 1.f;
    const float xv = off[i] >= 0 ? x[off[i]] : mean;
    g[i] = yv > 0.f ? gv : gv * slope;
    xh[i] = (xv - mean) * invstd;
  }
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    s += g[i];
    q += (double)g[i] * (double)xh[i];
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q, tot);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] = (float)tot[0];
    if (dgamma) dgamma[c] = (float)tot[1];
  }
  const double cnt = (double)N * HW;
  const float k0 = (gamma ? gamma[c] : 1.f) * invstd;
  const float mg = (float)(tot[0] / cnt), mgx = (float)(tot[1] / cnt);
#pragma unroll
  for (int i = 0; i < E; ++i) {
    if (off[i] < 0) continue;
    dx[off[i]] = k0 * (g[i] - mg - xh[i] * mgx);
    if (dres) dres[off[i]] = g[i];
  }
}

struct BnPeer {
  PeerTable tab;
  int me, world;
  PeerSeq sq;                   // exchange number: absolute, or relative to the device word of a replayed capture (peer_comm.h)
  size_t chan_off;              // byte offset of the channel records inside a slot
  long long timeout_ticks;
  unsigned long long *status;
  unsigned *wait_ring;
};

__device__ __forceinline__ unsigned long long bn_peer_seq(const BnPeer &pc) { return peer_seq_value(pc.sq); }
__device__ __forceinline__ size_t bn_peer_chan(const BnPeer &pc) {
  return (size_t)(peer_seq_value(pc.sq) % PEER_SLOTS) * pc.sq.slot_bytes + pc.chan_off;
}

// Vectorised register-cached forms (round 5): HW % 4 == 0 (every map of the head), 16-byte loads and stores -- element group
// q = tid + i T covers the channel's values 4q .. 4q+3 (one (n, c) row: a group never straddles rows).  A thread keeps E4 groups:
// 5 x 16-byte loads in flight per tensor instead of 17 x 4-byte ones, and a quarter of the index arithmetic.
// PEER: the multi-rank form -- the channel's sums meet the other ranks' between the block sum and the apply (peer_chan_exchange)
template <int T, int E4, bool PEER = false>
__global__ __launch_bounds__(T) void k_bn2d_fwd_small_rc4(const float *__restrict__ x, const float *__restrict__ res,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         int N, int C, int HW, float eps, float momentum, float slope,
                                                         float *__restrict__ run_mean, float *__restrict__ run_var,
                                                         float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                         float *__restrict__ y, double *__restrict__ count_out, BnPeer pc) {
  const int c = blockIdx.x, groups = (N * HW) >> 2;
  float4 v[E4], r[E4];
  int off[E4];
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    const int q = threadIdx.x + i * T;
    const int e = q << 2;
    const int n = e / HW, k = e - n * HW;
    off[i] = q < groups ? (n * C + c) * HW + k : -1;
    v[i] = off[i] >= 0 ? *reinterpret_cast<const float4 *>(x + off[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    r[i] = (res && off[i] >= 0) ? *reinterpret_cast<const float4 *>(res + off[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  double s = 0.0, q2 = 0.0;
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    s += ((double)v[i].x + v[i].y) + ((double)v[i].z + v[i].w);
    q2 += ((double)v[i].x * v[i].x + (double)v[i].y * v[i].y) + ((double)v[i].z * v[i].z + (double)v[i].w * v[i].w);
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q2, tot);
  double ex[3] = {tot[0], tot[1], (double)N * HW};
  bool ok = true;
  if constexpr (PEER) {
    __shared__ double sh[4 + 4 * PEER_MAX_WORLD];
    ok = peer_chan_exchange(pc.tab, pc.me, pc.world, bn_peer_seq(pc), bn_peer_chan(pc), c, pc.timeout_ticks, pc.status, ex, sh, pc.wait_ring);
  }
  const double cnt = ex[2];
  const double m = ex[0] / cnt;
  double var = ex[1] / cnt - m * m;
  var = var > 0.0 ? var : 0.0;
  float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (!ok) mean = invstd = __builtin_nanf("");          // a peer never arrived: poison, never hang (status says who)
  if (threadIdx.x == 0) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (PEER && c == 0) count_out[0] = cnt;
    if (run_mean) {
      const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float a1 = g * invstd, a0 = b - mean * a1;
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    if (off[i] < 0) continue;
    float4 o;
    o.x = v[i].x * a1 + a0 + r[i].x;
    o.y = v[i].y * a1 + a0 + r[i].y;
    o.z = v[i].z * a1 + a0 + r[i].z;
    o.w = v[i].w * a1 + a0 + r[i].w;
    o.x = o.x > 0.f ? o.x : o.x * slope;
    o.y = o.y > 0.f ? o.y : o.y * slope;
    o.z = o.z > 0.f ? o.z : o.z * slope;
    o.w = o.w > 0.f ? o.w : o.w * slope;
    *reinterpret_cast<float4 *>(y + off[i]) = o;
  }
}

template <int T, int E4, bool PEER = false>
__global__ __launch_bounds__(T) void k_bn2d_bwd_small_rc4(const float *__restrict__ dy, const float *__restrict__ y,
                                                         const float *__restrict__ x, const float *__restrict__ gamma,
                                                         const float *__restrict__ save_mean,
                                                         const float *__restrict__ save_invstd, int N, int C, int HW,
                                                         float slope, int has_act, float *__restrict__ dx,
                                                         float *__restrict__ dres, float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta, const double *__restrict__ count_all,
                                                         BnPeer pc) {
  const int c = blockIdx.x, groups = (N * HW) >> 2;
  const float mean = save_mean[c], invstd = save_invstd[c];
  float4 g[E4], xh[E4];
  int off[E4];
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    const int q = threadIdx.x + i * T;
    const int e = q << 2;
    const int n = e / HW, k = e - n * HW;
    off[i] = q < groups ? (n * C + c) * HW + k : -1;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 gv = off[i] >= 0 ? *reinterpret_cast<const float4 *>(dy + off[i]) : z;
    const float4 yv = (has_act && off[i] >= 0) ? *reinterpret_cast<const float4 *>(y + off[i]) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 xv = off[i] >= 0 ? *reinterpret_cast<const float4 *>(x + off[i]) : make_float4(mean, mean, mean, mean);
    g[i].x = yv.x > 0.f ? gv.x : gv.x * slope;
    g[i].y = yv.y > 0.f ? gv.y : gv.y * slope;
    g[i].z = yv.z > 0.f ? gv.z : gv.z * slope;
    g[i].w = yv.w > 0.f ? gv.w : gv.w * slope;
    xh[i].x = (xv.x - mean) * invstd;
    xh[i].y = (xv.y - mean) * invstd;
    xh[i].z = (xv.z - mean) * invstd;
    xh[i].w = (xv.w - mean) * invstd;
  }
  double s = 0.0, q2 = 0.0;
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    s += ((double)g[i].x + g[i].y) + ((double)g[i].z + g[i].w);
    q2 += ((double)g[i].x * xh[i].x + (double)g[i].y * xh[i].y) + ((double)g[i].z * xh[i].z + (double)g[i].w * xh[i].w);
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q2, tot);
  if (threadIdx.x == 0) {          // the affine gradients are this rank's own sums (data parallel averages them later)
    if (dbeta) dbeta[c] = (float)tot[0];
    if (dgamma) dgamma[c] = (float)tot[1];
  }
  double ex[3] = {tot[0], tot[1], 0.0};
  bool ok = true;
  if constexpr (PEER) {
    __shared__ double sh[4 + 4 * PEER_MAX_WORLD];
    ok = peer_chan_exchange(pc.tab, pc.me, pc.world, bn_peer_seq(pc), bn_peer_chan(pc), c, pc.timeout_ticks, pc.status, ex, sh, pc.wait_ring);
  }
  const double cnt = PEER ? *count_all : (double)N * HW;      // element count over all ranks, as exchanged in the forward pass
  const float k0 = (gamma ? gamma[c] : 1.f) * invstd;
  float mg = (float)(ex[0] / cnt), mgx = (float)(ex[1] / cnt);
  if (!ok) mg = mgx = __builtin_nanf("");
#pragma unroll
  for (int i = 0; i < E4; ++i) {
    if (off[i] < 0) continue;
    float4 o;
    o.x = k0 * (g[i].x - mg - xh[i].x * mgx);
    o.y = k0 * (g[i].y - mg - xh[i].y * mgx);
    o.z = k0 * (g[i].z - mg - xh[i].z * mgx);
    o.w = k0 * (g[i].w - mg - xh[i].w * mgx);
    *reinterpret_cast<float4 *>(dx + off[i]) = o;
    if (dres) *reinterpret_cast<float4 *>(dres + off[i]) = g[i];
  }
}

// Multi-rank SyncBN of the register-cached maps in ONE launch per direction (round 5).  The workgroup that owns channel c
// publishes its sums in this rank's peer slice, meets the workgroups of channel c on the other ranks (peer_chan_exchange,
// peer_comm.h: per-channel flags, rank-ordered sums = identical bits on every rank) and applies from its registers --
// instead of statistics kernel -> exchange kernel -> apply kernel with a second pass over the activation.  Same arithmetic
// as the three-launch path (double sums, the count exchanged with them).

template <int T, int E>
__global__ __launch_bounds__(T) void k_bn2d_fwd_rc_peer(const float *__restrict__ x, const float *__restrict__ res,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int N, int C, int HW, float eps, float momentum, float slope,
                                                       float *__restrict__ run_mean, float *__restrict__ run_var,
                                                       float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                       double *__restrict__ count_out, float *__restrict__ y, BnPeer pc) {
  const int c = blockIdx.x, total = N * HW;
  float v[E], r[E];
  int64_t off[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int e = threadIdx.x + i * T;
    const int n = e / HW, k = e - n * HW;
    off[i] = e < total ? ((int64_t)n * C + c) * HW + k : -1;
    v[i] = off[i] >= 0 ? x[off[i]] : 0.f;
    r[i] = (res && off[i] >= 0) ? res[off[i]] : 0.f;
  }
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    s += v[i];
    q += (double)v[i] * v[i];
  }
  __shared__ double tot[2];
  __shared__ double sh[4 + 4 * PEER_MAX_WORLD];
  bn_block_sum2_t<T>(s, q, tot);
  double ex[3] = {tot[0], tot[1], (double)N * HW};
  const bool ok = peer_chan_exchange(pc.tab, pc.me, pc.world, bn_peer_seq(pc), bn_peer_chan(pc), c, pc.timeout_ticks, pc.status, ex, sh, pc.wait_ring);
  const double cnt = ex[2];
  const double m = ex[0] / cnt;
  double var = ex[1] / cnt - m * m;
  var = var > 0.0 ? var : 0.0;
  float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (!ok) mean = invstd = __builtin_nanf("");          // a peer never arrived: poison, never hang (status says who)
  if (threadIdx.x == 0) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (c == 0) count_out[0] = cnt;
    if (run_mean) {
      const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float a1 = g * invstd, a0 = b - mean * a1;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    if (off[i] < 0) continue;
    const float o = v[i] * a1 + a0 + r[i];
    y[off[i]] = o > 0.f ? o : o * slope;
  }
}

template <int T, int E>
__global__ __launch_bounds__(T) void k_bn2d_bwd_rc_peer(const float *__restrict__ dy, const float *__restrict__ y,
                                                       const float *__restrict__ x, const float *__restrict__ gamma,
                                                       const float *__restrict__ save_mean,
                                                       const float *__restrict__ save_invstd,
                                                       const double *__restrict__ count_all, int N, int C, int HW,
                                                       float slope, int has_act, float *__restrict__ dx,
                                                       float *__restrict__ dres, float *__restrict__ dgamma,
                                                       float *__restrict__ dbeta, BnPeer pc) {
  const int c = blockIdx.x, total = N * HW;
  const float mean = save_mean[c], invstd = save_invstd[c];
  float g[E], xh[E];
  int64_t off[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int e = threadIdx.x + i * T;
    const int n = e / HW, k = e - n * HW;
    off[i] = e < total ? ((int64_t)n * C + c) * HW + k : -1;
    const float gv = off[i] >= 0 ? dy[off[i]] : 0.f;
    const float yv = (has_act && off[i] >= 0) ? y[off[i]] : 1.f;
    const float xv = off[i] >= 0 ? x[off[i]] : mean;
    g[i] = yv > 0.f ? gv : gv * slope;
    xh[i] = (xv - mean) * invstd;
  }
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    s += g[i];
    q += (double)g[i] * (double)xh[i];
  }
  __shared__ double tot[2];
  __shared__ double sh[4 + 4 * PEER_MAX_WORLD];
  bn_block_sum2_t<T>(s, q, tot);
  if (threadIdx.x == 0) {          // the affine gradients are this rank's own sums (data parallel averages them later)
    if (dbeta) dbeta[c] = (float)tot[0];
    if (dgamma) dgamma[c] = (float)tot[1];
  }
  double ex[3] = {tot[0], tot[1], 0.0};
  const bool ok = peer_chan_exchange(pc.tab, pc.me, pc.world, bn_peer_seq(pc), bn_peer_chan(pc), c, pc.timeout_ticks, pc.status, ex, sh, pc.wait_ring);
  const double cnt = *count_all;        // element count over all ranks, as exchanged in the forward pass
  const float k0 = (gamma ? gamma[c] : 1.f) * invstd;
  float mg = (float)(ex[0] / cnt), mgx = (float)(ex[1] / cnt);
  if (!ok) mg = mgx = __builtin_nanf("");
#pragma unroll
  for (int i = 0; i < E; ++i) {
    if (off[i] < 0) continue;
    dx[off[i]] = k0 * (g[i] - mg - xh[i] * mgx);
    if (dres) dres[off[i]] = g[i];
  }
}

// multi-rank forms of the small maps: the channel's sums go straight into the tensor that is all-reduced (no slice
// partials, no finish launch)
template <int T>
__global__ __launch_bounds__(T) void k_bn2d_stats_small(const float *__restrict__ x, int N, int C, int HW,
                                                        double *__restrict__ stats) {
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int n = 0; n < N; ++n) {
    const float *row = x + ((int64_t)n * C + c) * HW;
    for (int k = threadIdx.x; k < HW; k += T) {
      const float v = row[k];
      s += v;
      q += (double)v * v;
    }
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q, tot);
  if (threadIdx.x == 0) {
    stats[c] = tot[0];
    stats[C + c] = tot[1];
    if (c == 0) stats[2 * C] = (double)N * HW;
  }
}

template <int T>
__global__ __launch_bounds__(T) void k_bn2d_bwd_reduce_small(const float *__restrict__ dy, const float *__restrict__ y,
                                                             const float *__restrict__ x,
                                                             const float *__restrict__ save_mean,
                                                             const float *__restrict__ save_invstd, int N, int C,
                                                             int HW, float slope, int has_act, double *__restrict__ red,
                                                             float *__restrict__ dgamma, float *__restrict__ dbeta) {
  const int c = blockIdx.x;
  const float mean = save_mean[c], invstd = save_invstd[c];
  double s = 0.0, q = 0.0;
  for (int n = 0; n < N; ++n) {
    const int64_t base = ((int64_t)n * C + c) * HW;
    for (int k = threadIdx.x; k < HW; k += T) {
      float g = dy[base + k];
      if (has_act) g = y[base + k] > 0.f ? g : g * slope;
      s += g;
      q += (double)g * (double)((x[base + k] - mean) * invstd);
    }
  }
  __shared__ double tot[2];
  bn_block_sum2_t<T>(s, q, tot);
  if (threadIdx.x == 0) {
    red[c] = tot[0];
    red[C + c] = tot[1];
    if (dbeta) dbeta[c] = (float)tot[0];
    if (dgamma) dgamma[c] = (float)tot[1];
  }
}

static int64_t bn_per_blk(int N, int C, int HW, int *S) {
  const int64_t total = (int64_t)N * HW;
  int s = 1024 / (C > 0 ? C : 1);
  if (s < 1) s = 1;
  const int64_t max_s = rslo_cdiv(total, 2048);
  if (s > max_s) s = (int)(max_s > 0 ? max_s : 1);
  *S = s;
  return rslo_cdiv(rslo_cdiv(total, s), 4) * 4;     /* slices start on 16-byte boundaries when HW % 4 == 0 */
}

extern "C" size_t rslo_bn2d_ws_bytes(int N, int C, int HW) {
  int S;
  bn_per_blk(N, C, HW, &S);
  return (size_t)C * S * 2 * sizeof(double);
}

extern "C" int rslo_bn2d_stats(const float *x, int N, int C, int HW, void *ws, size_t ws_bytes, int32_t *done,
                               double *stats, void *stream) {
  RSLO_CHECK_ARG(x && ws && done && stats && N >= 1 && C >= 1 && HW >= 1, "rslo_bn2d_stats: bad arguments");
  RSLO_CHECK_ARG(ws_bytes >= rslo_bn2d_ws_bytes(N, C, HW), "rslo_bn2d_stats: workspace too small");
  if ((int64_t)N * HW <= BN_SMALL_MAX) {
    if (HW <= 1024)
      hipLaunchKernelGGL((k_bn2d_stats_small<256>), dim3(C), dim3(256), 0, (hipStream_t)stream, x, N, C, HW, stats);
    else
      hipLaunchKernelGGL((k_bn2d_stats_small<1024>), dim3(C), dim3(1024), 0, (hipStream_t)stream, x, N, C, HW, stats);
    RSLO_CHECK_LAUNCH("k_bn2d_stats_small");
    return RSLO_OK;
  }
  int S;
  const int64_t per = bn_per_blk(N, C, HW, &S);
  hipLaunchKernelGGL(k_bn2d_stats, dim3(S, C), dim3(BN_THREADS), 0, (hipStream_t)stream, x, N, C, HW, per, (double *)ws,
                     (int *)done, stats);
  RSLO_CHECK_LAUNCH("k_bn2d_stats");
  hipLaunchKernelGGL(k_bn2d_finish, dim3((unsigned)rslo_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, (const double *)ws,
                     S, C, (double)((int64_t)N * HW), stats, (float *)nullptr, (float *)nullptr);
  RSLO_CHECK_LAUNCH("k_bn2d_finish");
  return RSLO_OK;
}

extern "C" int rslo_bn2d_apply(const float *x, const float *res, const double *stats, const float *gamma,
                               const float *beta, int N, int C, int HW, float eps, float momentum, float act_slope,
                               float *running_mean, float *running_var, float *save_mean, float *save_invstd, float *y,
                               void *stream) {
  RSLO_CHECK_ARG(x && stats && save_mean && save_invstd && y, "rslo_bn2d_apply: bad arguments");
  dim3 grid((unsigned)rslo_cdiv(rslo_cdiv(HW, 4), BN_THREADS), (unsigned)(N * C));
  hipLaunchKernelGGL(k_bn2d_apply, grid, dim3(BN_THREADS), 0, (hipStream_t)stream, x, res, stats, gamma, beta, N, C, HW,
                     eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y,
                     (const double *)nullptr, 0);
  RSLO_CHECK_LAUNCH("k_bn2d_apply");
  return RSLO_OK;
}

extern "C" int rslo_bn2d_fwd_local(const float *x, const float *res, const float *gamma, const float *beta, int N, int C,
                                   int HW, float eps, float momentum, float act_slope, float *running_mean,
                                   float *running_var, float *save_mean, float *save_invstd, float *y, void *ws,
                                   size_t ws_bytes, void *stream) {
  RSLO_CHECK_ARG(x && ws && save_mean && save_invstd && y && N >= 1 && C >= 1 && HW >= 1, "rslo_bn2d_fwd_local: bad arguments");
  RSLO_CHECK_ARG(ws_bytes >= rslo_bn2d_ws_bytes(N, C, HW), "rslo_bn2d_fwd_local: workspace too small");
  if ((int64_t)N * HW <= BN_SMALL_MAX) {
    const int rc = rslo_tune(RSLO_TUNE_BN_SMALL_RC);      // 0: the two-pass loops; 2: the scalar register-cached forms (round 4)
    const int64_t groups = ((int64_t)N * HW) >> 2;
    const bool vec = rc == 1 && (HW & 3) == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0 &&
                     (int64_t)N * C * HW < ((int64_t)1 << 31);
    if (vec && groups <= 256 * 2)
      hipLaunchKernelGGL((k_bn2d_fwd_small_rc4<256, 2>), dim3(C), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, N, C,
                         HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y, (double *)nullptr, BnPeer{});
    else if (vec && groups <= 1024 * 2)
      hipLaunchKernelGGL((k_bn2d_fwd_small_rc4<1024, 2>), dim3(C), dim3(1024), 0, (hipStream_t)stream, x, res, gamma, beta, N, C,
                         HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y, (double *)nullptr, BnPeer{});
    else if (vec && groups <= 1024 * 5)
      hipLaunchKernelGGL((k_bn2d_fwd_small_rc4<1024, 5>), dim3(C), dim3(1024), 0, (hipStream_t)stream, x, res, gamma, beta, N, C,
                         HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y, (double *)nullptr, BnPeer{});
    else if (rc && (int64_t)N * HW <= 256 * BN_RC)
      hipLaunchKernelGGL((k_bn2d_fwd_small_rc<256>), dim3(C), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, N, C,
                         HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y);
    else if (rc && (int64_t)N * HW <= 1024 * BN_RC)
      hipLaunchKernelGGL((k_bn2d_fwd_small_rc<1024>), dim3(C), dim3(1024), 0, (hipStream_t)stream, x, res, gamma, beta, N, C,
                         HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y);
    else if (rc && (int64_t)N * HW <= 1024 * BN_RC_BIG)
      hipLaunchKernelGGL((k_bn2d_fwd_small_rc<1024, BN_RC_BIG>), dim3(C), dim3(1024), 0, (hipStream_t)stream, x, res, gamma,
                         beta, N, C, HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y);
    else if (HW <= 1024)
      hipLaunchKernelGGL((k_bn2d_fwd_small<256>), dim3(C), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, N, C, HW,
                         eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y);
    else
      hipLaunchKernelGGL((k_bn2d_fwd_small<1024>), dim3(C), dim3(1024), 0, (hipStream_t)stream, x, res, gamma, beta, N, C,
                         HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y);
    RSLO_CHECK_LAUNCH("k_bn2d_fwd_small");
    return RSLO_OK;
  }
  int S;
  const int64_t per = bn_per_blk(N, C, HW, &S);
  hipLaunchKernelGGL(k_bn2d_stats, dim3(S, C), dim3(BN_THREADS), 0, (hipStream_t)stream, x, N, C, HW, per, (double *)ws,
                     (int *)nullptr, (double *)nullptr);
  RSLO_CHECK_LAUNCH("k_bn2d_stats");
  dim3 grid((unsigned)rslo_cdiv(rslo_cdiv(HW, 4), BN_THREADS), (unsigned)(N * C));
  hipLaunchKernelGGL(k_bn2d_apply, grid, dim3(BN_THREADS), 0, (hipStream_t)stream, x, res, (const double *)nullptr, gamma,
                     beta, N, C, HW, eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y,
                     (const double *)ws, S);
  RSLO_CHECK_LAUNCH("k_bn2d_apply");
  return RSLO_OK;
}

extern "C" int rslo_bn2d_bwd_local(const float *dy, const float *y, const float *x, const float *gamma,
                                   const float *save_mean, const float *save_invstd, int N, int C, int HW,
                                   float act_slope, int has_act, float *dx, float *dres, float *dgamma, float *dbeta,
                                   void *ws, size_t ws_bytes, void *stream) {
  RSLO_CHECK_ARG(dy && x && save_mean && save_invstd && ws && dx, "rslo_bn2d_bwd_local: bad arguments");
  RSLO_CHECK_ARG(!has_act || y, "rslo_bn2d_bwd_local: y is needed for the activation mask");
  RSLO_CHECK_ARG(ws_bytes >= rslo_bn2d_ws_bytes(N, C, HW), "rslo_bn2d_bwd_local: workspace too small");
  if ((int64_t)N * HW <= BN_SMALL_MAX) {
    const int rc = rslo_tune(RSLO_TUNE_BN_SMALL_RC);
    const int64_t groups = ((int64_t)N * HW) >> 2;
    const bool vec = rc == 1 && (HW & 3) == 0 && (int64_t)N * C * HW < ((int64_t)1 << 31) &&
                     (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0;
    if (vec && groups <= 256 * 2)
      hipLaunchKernelGGL((k_bn2d_bwd_small_rc4<256, 2>), dim3(C), dim3(256), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta, (const double *)nullptr, BnPeer{});
    else if (vec && groups <= 1024 * 2)
      hipLaunchKernelGGL((k_bn2d_bwd_small_rc4<1024, 2>), dim3(C), dim3(1024), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta, (const double *)nullptr, BnPeer{});
    else if (vec && groups <= 1024 * 5)
      hipLaunchKernelGGL((k_bn2d_bwd_small_rc4<1024, 5>), dim3(C), dim3(1024), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta, (const double *)nullptr, BnPeer{});
    else if (rc && (int64_t)N * HW <= 256 * BN_RC)
      hipLaunchKernelGGL((k_bn2d_bwd_small_rc<256>), dim3(C), dim3(256), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta);
    else if (rc && (int64_t)N * HW <= 1024 * BN_RC)
      hipLaunchKernelGGL((k_bn2d_bwd_small_rc<1024>), dim3(C), dim3(1024), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta);
    else if (rc && (int64_t)N * HW <= 1024 * BN_RC_BIG)
      hipLaunchKernelGGL((k_bn2d_bwd_small_rc<1024, BN_RC_BIG>), dim3(C), dim3(1024), 0, (hipStream_t)stream, dy, y, x, gamma,
                         save_mean, save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta);
    else if (HW <= 1024)
      hipLaunchKernelGGL((k_bn2d_bwd_small<256>), dim3(C), dim3(256), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta);
    else
      hipLaunchKernelGGL((k_bn2d_bwd_small<1024>), dim3(C), dim3(1024), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta);
    RSLO_CHECK_LAUNCH("k_bn2d_bwd_small");
    return RSLO_OK;
  }
  int S;
  const int64_t per = bn_per_blk(N, C, HW, &S);
  hipLaunchKernelGGL(k_bn2d_bwd_reduce, dim3(S, C), dim3(BN_THREADS), 0, (hipStream_t)stream, dy, y, x, save_mean,
                     save_invstd, N, C, HW, per, act_slope, has_act, (double *)ws, (int *)nullptr, (double *)nullptr,
                     (float *)nullptr, (float *)nullptr);
  RSLO_CHECK_LAUNCH("k_bn2d_bwd_reduce");
  dim3 grid((unsigned)rslo_cdiv(rslo_cdiv(HW, 4), BN_THREADS), (unsigned)(N * C));
  hipLaunchKernelGGL(k_bn2d_bwd_apply, grid, dim3(BN_THREADS), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                     save_invstd, (const double *)nullptr, (double)((int64_t)N * HW), (const double *)nullptr, C, HW,
                     act_slope, has_act, dx, dres,
                     (const double *)ws, S, dgamma, dbeta);
  RSLO_CHECK_LAUNCH("k_bn2d_bwd_apply");
  return RSLO_OK;
}

extern "C" int rslo_bn2d_bwd_reduce(const float *dy, const float *y, const float *x, const float *save_mean,
                                    const float *save_invstd, int N, int C, int HW, float act_slope, int has_act,
                                    void *ws, size_t ws_bytes, int32_t *done, double *red, float *dgamma, float *dbeta,
                                    void *stream) {
  RSLO_CHECK_ARG(dy && x && save_mean && save_invstd && ws && done && red, "rslo_bn2d_bwd_reduce: bad arguments");
  RSLO_CHECK_ARG(!has_act || y, "rslo_bn2d_bwd_reduce: y is needed for the activation mask");
  RSLO_CHECK_ARG(ws_bytes >= rslo_bn2d_ws_bytes(N, C, HW), "rslo_bn2d_bwd_reduce: workspace too small");
  if ((int64_t)N * HW <= BN_SMALL_MAX) {
    if (HW <= 1024)
      hipLaunchKernelGGL((k_bn2d_bwd_reduce_small<256>), dim3(C), dim3(256), 0, (hipStream_t)stream, dy, y, x, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, red, dgamma, dbeta);
    else
      hipLaunchKernelGGL((k_bn2d_bwd_reduce_small<1024>), dim3(C), dim3(1024), 0, (hipStream_t)stream, dy, y, x, save_mean,
                         save_invstd, N, C, HW, act_slope, has_act, red, dgamma, dbeta);
    RSLO_CHECK_LAUNCH("k_bn2d_bwd_reduce_small");
    return RSLO_OK;
  }
  int S;
  const int64_t per = bn_per_blk(N, C, HW, &S);
  hipLaunchKernelGGL(k_bn2d_bwd_reduce, dim3(S, C), dim3(BN_THREADS), 0, (hipStream_t)stream, dy, y, x, save_mean,
                     save_invstd, N, C, HW, per, act_slope, has_act, (double *)ws, (int *)done, red, dgamma, dbeta);
  RSLO_CHECK_LAUNCH("k_bn2d_bwd_reduce");
  hipLaunchKernelGGL(k_bn2d_finish, dim3((unsigned)rslo_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, (const double *)ws,
                     S, C, -1.0, red, dbeta, dgamma);
  RSLO_CHECK_LAUNCH("k_bn2d_finish");
  return RSLO_OK;
}

extern "C" int rslo_bn2d_bwd_apply(const float *dy, const float *y, const float *x, const float *gamma,
                                   const float *save_mean, const float *save_invstd, const double *red, double count,
                                   const double *count_dev, int N, int C, int HW, float act_slope, int has_act,
                                   float *dx, float *dres, void *stream) {
  RSLO_CHECK_ARG(dy && x && save_mean && save_invstd && red && dx && (count > 0 || count_dev),
                 "rslo_bn2d_bwd_apply: bad arguments");
  dim3 grid((unsigned)rslo_cdiv(rslo_cdiv(HW, 4), BN_THREADS), (unsigned)(N * C));
  hipLaunchKernelGGL(k_bn2d_bwd_apply, grid, dim3(BN_THREADS), 0, (hipStream_t)stream, dy, y, x, gamma, save_mean,
                     save_invstd, red, count, count_dev, C, HW, act_slope, has_act, dx, dres, (const double *)nullptr, 0,
                     (float *)nullptr, (float *)nullptr);
  RSLO_CHECK_LAUNCH("k_bn2d_bwd_apply");
  return RSLO_OK;
}

// ---- multi-rank single-launch entry points (register-cached maps) ---------------------------------------------------------
// 1 when rslo_bn2d_fwd_peer / _bwd_peer take the shape (else the caller runs statistics -> rslo_peer_allreduce_f64 -> apply)
extern "C" int rslo_bn2d_peer_supported(int N, int C, int HW) {
  return rslo_tune(RSLO_TUNE_BN_SMALL_RC) && C >= 1 && C <= PEER_MAX_CH && (int64_t)N * HW <= 1024 * BN_RC_BIG;
}

static BnPeer bn_peer_args(RsloPeerComm *c) {
  BnPeer pc;
  pc.tab = c->tab; pc.me = c->rank; pc.world = c->world;
  pc.sq = peer_next_seq(c);
  pc.chan_off = peer_chan_off(c->max_n);
  pc.timeout_ticks = c->timeout_ticks; pc.status = c->status_dev;
  pc.wait_ring = c->wait_ring_dev;
  // (host write to pinned memory, ordered before the launch that follows; a captured launch: rslo_peer_replay_prepare)
  if (!c->capturing) c->wait_ring_host[pc.sq.seq % PEER_WAIT_RING] = 0;
  return pc;
}

extern "C" int rslo_bn2d_fwd_peer(void *comm, const float *x, const float *res, const float *gamma, const float *beta, int N,
                                  int C, int HW, float eps, float momentum, float act_slope, float *running_mean,
                                  float *running_var, float *save_mean, float *save_invstd, double *count_out, float *y,
                                  void *stream) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && x && save_mean && save_invstd && count_out && y && rslo_bn2d_peer_supported(N, C, HW),
                 "rslo_bn2d_fwd_peer: bad arguments / shape outside the single-launch range (N*HW = %lld, C = %d)",
                 (long long)N * HW, C);
  const BnPeer pc = bn_peer_args(c);
  const int64_t per = (int64_t)N * HW;
  hipStream_t st = (hipStream_t)stream;
#define BN_GO(T, E) hipLaunchKernelGGL((k_bn2d_fwd_rc_peer<T, E>), dim3(C), dim3(T), 0, st, x, res, gamma, beta, N, C, HW, eps, \
                                       momentum, act_slope, running_mean, running_var, save_mean, save_invstd, count_out, y, pc)
#define BN_GO4(T, E) hipLaunchKernelGGL((k_bn2d_fwd_small_rc4<T, E, true>), dim3(C), dim3(T), 0, st, x, res, gamma, beta, N, C, HW, \
                                        eps, momentum, act_slope, running_mean, running_var, save_mean, save_invstd, y, count_out, pc)
  const int64_t groups = per >> 2;
  const bool vec = rslo_tune(RSLO_TUNE_BN_SMALL_RC) == 1 && (HW & 3) == 0 && (int64_t)N * C * HW < ((int64_t)1 << 31) &&
                   (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0;
  if (vec && groups <= 256 * 2) BN_GO4(256, 2);
  else if (vec && groups <= 1024 * 2) BN_GO4(1024, 2);
  else if (vec && groups <= 1024 * 5) BN_GO4(1024, 5);
  else if (per <= 256 * BN_RC) BN_GO(256, BN_RC);
  else if (per <= 1024 * BN_RC) BN_GO(1024, BN_RC);
  else BN_GO(1024, BN_RC_BIG);
#undef BN_GO
#undef BN_GO4
  RSLO_CHECK_LAUNCH("k_bn2d_fwd_rc_peer");
  peer_commit_seq(c);
  return RSLO_OK;
}

extern "C" int rslo_bn2d_bwd_peer(void *comm, const float *dy, const float *y, const float *x, const float *gamma,
                                  const float *save_mean, const float *save_invstd, const double *count_all, int N, int C,
                                  int HW, float act_slope, int has_act, float *dx, float *dres, float *dgamma, float *dbeta,
                                  void *stream) {
  RsloPeerComm *c = (RsloPeerComm *)comm;
  RSLO_CHECK_ARG(c && dy && x && save_mean && save_invstd && count_all && dx && rslo_bn2d_peer_supported(N, C, HW),
                 "rslo_bn2d_bwd_peer: bad arguments / shape outside the single-launch range");
  RSLO_CHECK_ARG(!has_act || y, "rslo_bn2d_bwd_peer: y is needed for the activation mask");
  const BnPeer pc = bn_peer_args(c);
  const int64_t per = (int64_t)N * HW;
  hipStream_t st = (hipStream_t)stream;
#define BN_GO(T, E) hipLaunchKernelGGL((k_bn2d_bwd_rc_peer<T, E>), dim3(C), dim3(T), 0, st, dy, y, x, gamma, save_mean, save_invstd, \
                                       count_all, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta, pc)
#define BN_GO4(T, E) hipLaunchKernelGGL((k_bn2d_bwd_small_rc4<T, E, true>), dim3(C), dim3(T), 0, st, dy, y, x, gamma, save_mean, \
                                        save_invstd, N, C, HW, act_slope, has_act, dx, dres, dgamma, dbeta, count_all, pc)
  const int64_t groups = per >> 2;
  const bool vec = rslo_tune(RSLO_TUNE_BN_SMALL_RC) == 1 && (HW & 3) == 0 && (int64_t)N * C * HW < ((int64_t)1 << 31) &&
                   (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0;
  if (vec && groups <= 256 * 2) BN_GO4(256, 2);
  else if (vec && groups <= 1024 * 2) BN_GO4(1024, 2);
  else if (vec && groups <= 1024 * 5) BN_GO4(1024, 5);
  else if (per <= 256 * BN_RC) BN_GO(256, BN_RC);
  else if (per <= 1024 * BN_RC) BN_GO(1024, BN_RC);
  else BN_GO(1024, BN_RC_BIG);
#undef BN_GO
#undef BN_GO4
  RSLO_CHECK_LAUNCH("k_bn2d_bwd_rc_peer");
  peer_commit_seq(c);
  return RSLO_OK;
}
