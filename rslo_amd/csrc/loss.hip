// Fused kernels of the self-supervised consistency loss for gfx950.
//
//  * covariance-weighted residual, forward and backward: one thread per correspondence rebuilds both
//    3x3 covariances from their 7 parameters (cumulative eigenvalues + quaternion), gathers the partner by
//    the chamfer index, forms sigma = S1 + R S2 R^T, its closed-form inverse / determinant and the
//    Mahalanobis + log-det terms; block partial sums in double, deterministic second-stage reduce.
//    The backward recomputes instead of storing and scatter-adds the partner gradients with atomics
//    (the reference's own CUDA backward uses atomicAdd too, chamfer_distance.cu:177-206).
//  * ICP step: weighted-Kabsch moments of the ROI in double (one pass, raw moments), then a single
//    thread does the 3x3 Jacobi SVD, the reflection fix and composes the running (R, t) -- no host
//    round trip for torch.svd / det (rslo/layers/svd.py:36-46).
// All of it is latency/HBM-bound elementwise work on < 3 MB per pair.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "rslo_common.h"

#define LS_THREADS 256

struct Cov7 {
  float lam[3];
  float qn;      // |q|
  float qh[4];   // q / (|q| + 1e-9)
  float qhn;     // max(|qh|, 1e-12)
  float qt[4];   // qh / qhn  (x, y, z, w)
  float V[9];    // rotation matrix of qt, row-major
};

__device__ __forceinline__ void cov_build(const float *__restrict__ p, Cov7 &c, float S[9]) {
  c.lam[0] = p[0];
  c.lam[1] = c.lam[0] + p[1];
  c.lam[2] = c.lam[1] + p[2];
  const float n = sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
  c.qn = n;
  const float inv = 1.0f / (n + 1e-9f);
#pragma unroll
  for (int k = 0; k < 4; ++k) c.qh[k] = p[3 + k] * inv;
  float hn = sqrtf(c.qh[0] * c.qh[0] + c.qh[1] * c.qh[1] + c.qh[2] * c.qh[2] + c.qh[3] * c.qh[3]);
  hn = fmaxf(hn, 1e-12f);
  c.qhn = hn;
#pragma unroll
  for (int k = 0; k < 4; ++k) c.qt[k] = c.qh[k] / hn;
  const float x = c.qt[0], y = c.qt[1], z = c.qt[2], w = c.qt[3];
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  float *V = c.V;
  V[0] = 1.f - (tyy + tzz); V[1] = txy - twz;         V[2] = txz + twy;
  V[3] = txy + twz;         V[4] = 1.f - (txx + tzz); V[5] = tyz - twx;
  V[6] = txz - twy;         V[7] = tyz + twx;         V[8] = 1.f - (txx + tyy);
  // S = V diag(lam) V^T
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      S[a * 3 + b] = V[a * 3 + 0] * c.lam[0] * V[b * 3 + 0] + V[a * 3 + 1] * c.lam[1] * V[b * 3 + 1] +
                     V[a * 3 + 2] * c.lam[2] * V[b * 3 + 2];
}

// gradient of the 7 parameters given G = dL/dS (symmetric)
__device__ __forceinline__ void cov_backward(const float *__restrict__ p, const Cov7 &c, const float G[9],
                                             float g[7]) {
  const float *V = c.V;
  float GV[9];   // G V
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k) GV[a * 3 + k] = G[a * 3 + 0] * V[0 * 3 + k] + G[a * 3 + 1] * V[1 * 3 + k] + G[a * 3 + 2] * V[2 * 3 + k];
  float dl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) dl[k] = V[0 * 3 + k] * GV[0 * 3 + k] + V[1 * 3 + k] * GV[1 * 3 + k] + V[2 * 3 + k] * GV[2 * 3 + k];
  g[0] = dl[0] + dl[1] + dl[2];
  g[1] = dl[1] + dl[2];
  g[2] = dl[2];
  float dV[9];   // 2 G V diag(lam)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k) dV[a * 3 + k] = 2.f * c.lam[k] * GV[a * 3 + k];
  const float x = c.qt[0], y = c.qt[1], z = c.qt[2], w = c.qt[3];
  float dq[4];
  dq[0] = dV[1] * 2 * y + dV[2] * 2 * z + dV[3] * 2 * y - dV[4] * 4 * x - dV[5] * 2 * w + dV[6] * 2 * z + dV[7] * 2 * w - dV[8] * 4 * x;
  dq[1] = -dV[0] * 4 * y + dV[1] * 2 * x + dV[2] * 2 * w + dV[3] * 2 * x + dV[5] * 2 * z - dV[6] * 2 * w + dV[7] * 2 * z - dV[8] * 4 * y;
  dq[2] = -dV[0] * 4 * z - dV[1] * 2 * w + dV[2] * 2 * x + dV[3] * 2 * w - dV[4] * 4 * z + dV[5] * 2 * y + dV[6] * 2 * x + dV[7] * 2 * y;
  dq[3] = -dV[1] * 2 * z + dV[2] * 2 * y + dV[3] * 2 * z - dV[5] * 2 * x - dV[6] * 2 * y + dV[7] * 2 * x;
  // qt = qh / max(|qh|, eps)
  float dot = c.qt[0] * dq[0] + c.qt[1] * dq[1] + c.qt[2] * dq[2] + c.qt[3] * dq[3];
  float dh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) dh[k] = (dq[k] - c.qt[k] * dot) / c.qhn;
  // qh = q / (|q| + 1e-9)
  const float n = c.qn, ne = n + 1e-9f;
  float qd = p[3] * dh[0] + p[4] * dh[1] + p[5] * dh[2] + p[6] * dh[3];
  const float coef = (n > 0.f) ? qd / (n * ne * ne) : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) g[3 + k] = dh[k] / ne - p[3 + k] * coef;
}

__device__ __forceinline__ float inv3(const float S[9], float I[9]) {
  const float a = S[0], b = S[1], c = S[2], d = S[3], e = S[4], f = S[5], g = S[6], h = S[7], i = S[8];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + b * B + c * C;
  const float r = 1.0f / det;
  I[0] = A * r;  I[1] = -(b * i - c * h) * r;  I[2] = (b * f - c * e) * r;
  I[3] = B * r;  I[4] = (a * i - c * g) * r;   I[5] = -(a * f - c * d) * r;
  I[6] = C * r;  I[7] = -(a * h - b * g) * r;  I[8] = (a * e - b * d) * r;
  return det;
}

// sigma = S1 + R S2 R^T
__device__ __forceinline__ void sigma_of(const float S1[9], const float S2[9], const float R[9], float sg[9]) {
  float RS[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) RS[a * 3 + b] = R[a * 3 + 0] * S2[0 * 3 + b] + R[a * 3 + 1] * S2[1 * 3 + b] + R[a * 3 + 2] * S2[2 * 3 + b];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      sg[a * 3 + b] = S1[a * 3 + b] + RS[a * 3 + 0] * R[b * 3 + 0] + RS[a * 3 + 1] * R[b * 3 + 1] + RS[a * 3 + 2] * R[b * 3 + 2];
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < LS_THREADS / 64; ++w) s += sh[w];
  return s;   // valid on thread 0
}

// ---------------------------------------------------------------------------------------
// residual forward: partial[b][block][3] = sum roi*sq, sum roi*0.5*log det, sum roi
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(LS_THREADS) void k_resid_fwd(const float *__restrict__ p1, const float *__restrict__ tgt,
                                                          const float *__restrict__ cov1, const float *__restrict__ cov2,
                                                          const int32_t *__restrict__ idx, const float *__restrict__ dist,
                                                          const float *__restrict__ thr, const float *__restrict__ Rd,
                                                          int N, int M, double *__restrict__ partial) {
  __shared__ double sh[LS_THREADS / 64];
  const int b = blockIdx.y;
  const int i = blockIdx.x * LS_THREADS + threadIdx.x;
  double sq = 0.0, ld = 0.0, cnt = 0.0;
  if (i < N && dist[(int64_t)b * N + i] < thr[b]) {
    const int j = idx[(int64_t)b * N + i];
    const float *a = p1 + ((int64_t)b * N + i) * 3, *t = tgt + ((int64_t)b * M + j) * 3;
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rd[b * 9 + k];
    Cov7 c1, c2;
    float S1[9], S2[9], sg[9], I[9];
    cov_build(cov1 + ((int64_t)b * N + i) * 7, c1, S1);
    cov_build(cov2 + ((int64_t)b * M + j) * 7, c2, S2);
    sigma_of(S1, S2, R, sg);
    const float det = inv3(sg, I);
    const float d0 = a[0] - t[0], d1 = a[1] - t[1], d2 = a[2] - t[2];
    const float v0 = I[0] * d0 + I[1] * d1 + I[2] * d2, v1 = I[3] * d0 + I[4] * d1 + I[5] * d2,
                v2 = I[6] * d0 + I[7] * d1 + I[8] * d2;
    sq = (double)(d0 * v0 + d1 * v1 + d2 * v2);
    ld = (double)(0.5f * logf(det));
    cnt = 1.0;
  }
  const double s0 = block_sum(sq, sh), s1 = block_sum(ld, sh), s2 = block_sum(cnt, sh);
  if (threadIdx.x == 0) {
    double *o = partial + ((int64_t)b * gridDim.x + blockIdx.x) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}

__global__ void k_resid_finish(const double *__restrict__ partial, int nblk, float reg, float *__restrict__ loss,
                               float *__restrict__ cnt_out) {
  // the block partials come in through LDS, all loads of a chunk in flight together; thread 0 then adds them in block
  // order as before (one thread walking nblk x 3 dependent global loads was 19 us of a serial stretch of the step)
  __shared__ double sh[64 * 3];
  const int b = blockIdx.x;
  double s0 = 0, s1 = 0, s2 = 0;
  for (int k0 = 0; k0 < nblk; k0 += 64) {
    const int nk = nblk - k0 < 64 ? nblk - k0 : 64;
    for (int i = threadIdx.x; i < nk * 3; i += blockDim.x) sh[i] = partial[((int64_t)b * nblk + k0) * 3 + i];
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < nk; ++k) {
        s0 += sh[k * 3 + 0]; s1 += sh[k * 3 + 1]; s2 += sh[k * 3 + 2];
      }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  loss[b] = (float)(s0 / s2 + (double)reg * (s1 / s2));
  cnt_out[b] = (float)s2;
}

extern "C" size_t rslo_cov_residual_ws_bytes(int B, int N) {
  return (size_t)(B > 0 ? B : 1) * (size_t)rslo_cdiv(N > 0 ? N : 1, LS_THREADS) * 3 * sizeof(double);
}

extern "C" int rslo_cov_residual_fwd(const float *p1, const float *tgt, const float *cov1, const float *cov2,
                                              const int32_t *idx, const float *dist, const float *thr,
                                              const float *Rd, int B, int N, int M, float reg_weight, void *ws,
                                              size_t ws_bytes, float *loss, float *cnt, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return RSLO_OK;
  RSLO_CHECK_ARG(N >= 1 && M >= 1, "cov_residual_fwd: empty clouds");
  if (ws_bytes < rslo_cov_residual_ws_bytes(B, N)) {
    rslo_set_error("cov_residual_fwd: workspace too small");
    return RSLO_EWS;
  }
  const int nblk = (int)rslo_cdiv(N, LS_THREADS);
  hipLaunchKernelGGL(k_resid_fwd, dim3(nblk, B), dim3(LS_THREADS), 0, st, p1, tgt, cov1, cov2, idx, dist, thr, Rd, N,
                     M, (double *)ws);
  hipLaunchKernelGGL(k_resid_finish, dim3(B), dim3(64), 0, st, (const double *)ws, nblk, reg_weight, loss, cnt);
  RSLO_CHECK_LAUNCH("cov_residual_fwd");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// residual backward: gloss[b] -> gtgt (scatter), gcov1 (direct), gcov2 (scatter)
// ---------------------------------------------------------------------------------------
#define RB_THREADS 64
#define RB_NV 10              // values a source adds to its partner row: 3 (target point) + 7 (covariance parameters)
#define RB_SRC_BITS 32        // key = partner row (b M + j) << 32 | source row (b N + i): each half is one dword of the key
#define RB_SHORT 32           // segments up to this length are added by one thread, longer ones by one wave
__global__ __launch_bounds__(RB_THREADS) void k_resid_bwd(const float *__restrict__ p1, const float *__restrict__ tgt,
                                                          const float *__restrict__ cov1, const float *__restrict__ cov2,
                                                          const int32_t *__restrict__ idx, const float *__restrict__ dist,
                                                          const float *__restrict__ thr, const float *__restrict__ Rd,
                                                          const float *__restrict__ gloss, const float *__restrict__ cnt,
                                                          int N, int M, float reg, float *__restrict__ gp1,
                                                          float *__restrict__ gtgt, float *__restrict__ gcov1,
                                                          float *__restrict__ gcov2, float *__restrict__ contrib,
                                                          unsigned long long *__restrict__ keys, int *__restrict__ n_long) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * RB_THREADS + threadIdx.x;
  const int lane = threadIdx.x;                      // RB_THREADS == 64: one wave per block
  const bool in_range = i < N;
  // the partner rows no source reaches stay zero and the long-run counter starts at zero -- written here (the grid covers
  // max(N, M) rows), not by three fill launches in front of this kernel; the gather kernels run behind it
  if (i < M) {
    float *gt = gtgt + ((int64_t)b * M + i) * 3, *g2 = gcov2 + ((int64_t)b * M + i) * 7;
    gt[0] = gt[1] = gt[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) g2[k] = 0.f;
  }
  if (b == 0 && i == 0) *n_long = 0;
  const bool roi = in_range && (dist[(int64_t)b * N + i] < thr[b]);
  // values this lane scatter-adds to its partner row j: 3 (target point) + 7 (target covariance parameters)
  float sc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) sc[k] = 0.f;
  int j = -1 - lane;                                 // unique key for lanes without a partner
  if (in_range && !roi) {
    float *g1 = gcov1 + ((int64_t)b * N + i) * 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) g1[k] = 0.f;
    if (gp1) {
      gp1[((int64_t)b * N + i) * 3 + 0] = 0.f;
      gp1[((int64_t)b * N + i) * 3 + 1] = 0.f;
      gp1[((int64_t)b * N + i) * 3 + 2] = 0.f;
    }
  }
  if (roi) {
    float *g1 = gcov1 + ((int64_t)b * N + i) * 7;
    const float wgt = gloss[b] / cnt[b];
    j = idx[(int64_t)b * N + i];
    const float *a = p1 + ((int64_t)b * N + i) * 3, *t = tgt + ((int64_t)b * M + j) * 3;
    const float *pc1 = cov1 + ((int64_t)b * N + i) * 7, *pc2 = cov2 + ((int64_t)b * M + j) * 7;
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rd[b * 9 + k];
    Cov7 c1, c2;
    float S1[9], S2[9], sg[9], I[9];
    cov_build(pc1, c1, S1);
    cov_build(pc2, c2, S2);
    sigma_of(S1, S2, R, sg);
    inv3(sg, I);
    const float d0 = a[0] - t[0], d1 = a[1] - t[1], d2 = a[2] - t[2];
    float v[3] = {I[0] * d0 + I[1] * d1 + I[2] * d2, I[3] * d0 + I[4] * d1 + I[5] * d2,
                  I[6] * d0 + I[7] * d1 + I[8] * d2};
    // sigma^-1 is symmetric up to rounding: use the symmetrised inverse for the log-det term
    float G[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        G[r * 3 + c] = wgt * (-v[r] * v[c] + reg * 0.25f * (I[r * 3 + c] + I[c * 3 + r]));
    // d(sq)/dd = (sigma^-1 + sigma^-T) d
    const float u0 = I[0] * d0 + I[3] * d1 + I[6] * d2, u1 = I[1] * d0 + I[4] * d1 + I[7] * d2,
                u2 = I[2] * d0 + I[5] * d1 + I[8] * d2;
    const float gd[3] = {wgt * (v[0] + u0), wgt * (v[1] + u1), wgt * (v[2] + u2)};
    if (gp1) {
      gp1[((int64_t)b * N + i) * 3 + 0] = gd[0];
      gp1[((int64_t)b * N + i) * 3 + 1] = gd[1];
      gp1[((int64_t)b * N + i) * 3 + 2] = gd[2];
    }
    sc[0] = -gd[0];
    sc[1] = -gd[1];
    sc[2] = -gd[2];
    float g7[7];
    cov_backward(pc1, c1, G, g7);
#pragma unroll
    for (int k = 0; k < 7; ++k) g1[k] = g7[k];
    // G2 = R^T G R
    float RtG[9], G2[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) RtG[r * 3 + c] = R[0 * 3 + r] * G[0 * 3 + c] + R[1 * 3 + r] * G[1 * 3 + c] + R[2 * 3 + r] * G[2 * 3 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) G2[r * 3 + c] = RtG[r * 3 + 0] * R[0 * 3 + c] + RtG[r * 3 + 1] * R[1 * 3 + c] + RtG[r * 3 + 2] * R[2 * 3 + c];
    cov_backward(pc2, c2, G2, g7);
#pragma unroll
    for (int k = 0; k < 7; ++k) sc[3 + k] = g7[k];
  }
  // Nothing is added here.  Every source leaves its ten partner values and the key (partner row, source row);
  // rslo_cov_residual_bwd sorts the keys and k_resid_gather* add each partner's values in ascending source order -- the same
  // bits whatever the scheduling (per-run atomics, rounds 1-4, made every training step unique; that path is deleted).
  if (in_range) {
    float *o = contrib + ((int64_t)b * N + i) * RB_NV;
#pragma unroll
    for (int k = 0; k < 10; ++k) o[k] = sc[k];
    keys[(int64_t)b * N + i] = roi ? ((((unsigned long long)b * M + j) << RB_SRC_BITS) | ((unsigned long long)b * N + i)) : ~0ull;
  }
}

// one thread per sorted key: the head of a partner's run adds the run's values in key (= source) order
__global__ __launch_bounds__(256) void k_resid_gather(const unsigned long long *__restrict__ skeys, int64_t n,
                                                      const float *__restrict__ contrib, float *__restrict__ gtgt,
                                                      float *__restrict__ gcov2, int *__restrict__ n_long,
                                                      int64_t *__restrict__ long_list) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const unsigned long long k = skeys[p];
  if (k == ~0ull) return;
  const unsigned long long row = k >> RB_SRC_BITS;
  if (p > 0 && (skeys[p - 1] >> RB_SRC_BITS) == row) return;      // not the head of its run
  int64_t e = p + 1;
  while (e < n && e - p <= RB_SHORT && (skeys[e] >> RB_SRC_BITS) == row) ++e;
  if (e - p > RB_SHORT) {                                          // a long run: one wave adds it (k_resid_gather_long)
    long_list[atomicAdd(n_long, 1)] = p;                           // (the list's order does not matter: disjoint outputs)
    return;
  }
  float acc[RB_NV];
#pragma unroll
  for (int q = 0; q < RB_NV; ++q) acc[q] = 0.f;
  for (int64_t t = p; t < e; ++t) {
    const float *c = contrib + (int64_t)(unsigned)skeys[t] * RB_NV;      // low dword = source row
#pragma unroll
    for (int q = 0; q < RB_NV; ++q) acc[q] += c[q];
  }
  float *gt = gtgt + (int64_t)row * 3, *g2 = gcov2 + (int64_t)row * 7;
  gt[0] = acc[0]; gt[1] = acc[1]; gt[2] = acc[2];
#pragma unroll
  for (int q = 0; q < 7; ++q) g2[q] = acc[3 + q];
}

// one wave per long run: lane l adds elements l, l + 64, ... in order, then the 64 lane sums are added in a fixed tree.
// (Four waves per run measured 34 -> ~15 us on the bench's random-init state, where a few partners collect thousands of
// sources -- not kept: another summation order is another weight trajectory, and the parity tests pin this one.)
__global__ __launch_bounds__(64) void k_resid_gather_long(const unsigned long long *__restrict__ skeys, int64_t n,
                                                          const float *__restrict__ contrib, float *__restrict__ gtgt,
                                                          float *__restrict__ gcov2, const int *__restrict__ n_long,
                                                          const int64_t *__restrict__ long_list) {
  const int lane = threadIdx.x;
  for (int s = blockIdx.x; s < *n_long; s += gridDim.x) {
    const int64_t p = long_list[s];
    const unsigned long long row = skeys[p] >> RB_SRC_BITS;
    float acc[RB_NV];
#pragma unroll
    for (int q = 0; q < RB_NV; ++q) acc[q] = 0.f;
    for (int64_t t = p + lane; t < n && (skeys[t] >> RB_SRC_BITS) == row; t += 64) {
      const float *c = contrib + (int64_t)(unsigned)skeys[t] * RB_NV;
#pragma unroll
      for (int q = 0; q < RB_NV; ++q) acc[q] += c[q];
    }
#pragma unroll
    for (int q = 0; q < RB_NV; ++q)
      for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 64);
    if (lane == 0) {
      float *gt = gtgt + (int64_t)row * 3, *g2 = gcov2 + (int64_t)row * 7;
      gt[0] = acc[0]; gt[1] = acc[1]; gt[2] = acc[2];
#pragma unroll
      for (int q = 0; q < 7; ++q) g2[q] = acc[3 + q];
    }
  }
}

static int resid_key_bits(int B, int M) {
  int bits = 1;
  while ((1ull << bits) <= (unsigned long long)B * (unsigned long long)M) ++bits;      // all-ones > every partner row
  return bits + RB_SRC_BITS;
}

static size_t resid_sort_tmp_bytes(int64_t n, int bits) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_keys(nullptr, tmp, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n,
                                 (unsigned)RB_SRC_BITS, (unsigned)bits, (hipStream_t)0);
  return (tmp + 255) / 256 * 256;
}

// workspace of the ordered scatter: contributions [B N][10] f32 | keys [B N] u64 | sorted keys | long-run list [B N / 32 + 1]
// i64 | counter | the sort's own scratch
extern "C" size_t rslo_cov_residual_bwd_ws_bytes(int B, int N, int M) {
  if (B <= 0 || N <= 0 || M <= 0) return 256;
  const int64_t n = (int64_t)B * N;
  return (size_t)n * RB_NV * 4 + (size_t)n * 8 * 2 + (size_t)(n / RB_SHORT + 2) * 8 + 256 + resid_sort_tmp_bytes(n, resid_key_bits(B, M)) +
         8 * 256;      // + alignment of the six sections
}

extern "C" int rslo_cov_residual_bwd(const float *p1, const float *tgt, const float *cov1, const float *cov2,
                                              const int32_t *idx, const float *dist, const float *thr,
                                              const float *Rd, const float *gloss, const float *cnt, int B, int N,
                                              int M, float reg_weight, float *gp1 /*or NULL*/, float *gtgt,
                                              float *gcov1, float *gcov2, void *ws, size_t ws_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return RSLO_OK;
  const int64_t n = (int64_t)B * N;
  RSLO_CHECK_ARG(n < (1ll << 31) && (int64_t)B * M < (1ll << 31), "cov_residual_bwd: more than 2^31 rows");
  RSLO_CHECK_ARG(ws && ws_bytes >= rslo_cov_residual_bwd_ws_bytes(B, N, M), "cov_residual_bwd: workspace too small");
  const int bits = resid_key_bits(B, M);
  unsigned char *w = (unsigned char *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);      // every section on a 256-byte boundary
  auto take = [&](size_t bytes) { unsigned char *p = w; w += (bytes + 255) / 256 * 256; return p; };
  float *contrib = (float *)take((size_t)n * RB_NV * 4);
  unsigned long long *keys = (unsigned long long *)take((size_t)n * 8);
  unsigned long long *skeys = (unsigned long long *)take((size_t)n * 8);
  int64_t *long_list = (int64_t *)take((size_t)(n / RB_SHORT + 2) * 8);
  int *n_long = (int *)take(256);
  size_t tmp = resid_sort_tmp_bytes(n, bits);
  hipLaunchKernelGGL(k_resid_bwd, dim3((unsigned)rslo_cdiv(N > M ? N : M, RB_THREADS), B), dim3(RB_THREADS), 0, st, p1, tgt,
                     cov1, cov2, idx, dist, thr, Rd, gloss, cnt, N, M, reg_weight, gp1, gtgt, gcov1, gcov2, contrib, keys,
                     n_long);
  RSLO_CHECK_LAUNCH("cov_residual_bwd");
  // The keys are written in source order (key p belongs to source row p), and the radix sort is stable: sorting on the
  // partner half alone leaves every partner's run in ascending source order -- the order the full 49-bit sort gives, in 3
  // digit passes instead of 7.
  RSLO_HIP(rocprim::radix_sort_keys((void *)w, tmp, keys, skeys, (size_t)n, (unsigned)RB_SRC_BITS, (unsigned)bits, st));
  hipLaunchKernelGGL(k_resid_gather, dim3((unsigned)rslo_cdiv(n, 256)), dim3(256), 0, st, (const unsigned long long *)skeys, n,
                     (const float *)contrib, gtgt, gcov2, n_long, long_list);
  hipLaunchKernelGGL(k_resid_gather_long, dim3(512), dim3(64), 0, st, (const unsigned long long *)skeys, n,
                     (const float *)contrib, gtgt, gcov2, (const int *)n_long, (const int64_t *)long_list);
  RSLO_CHECK_LAUNCH("cov_residual_bwd(gather)");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// ICP step (rslo/core/losses.py:449-488 + rslo/layers/svd.py:14-64)
// moments (23 doubles): m, m*s[3], m*t[3], mw, mw*s[3], mw*t[3], mw*s(x)t[9]
// with m = roi (dist < thr), w = cos(n1, assoc - p1)^2, s = p1, t = tgt[idx]
// ---------------------------------------------------------------------------------------
#define ICP_NM 23

__global__ __launch_bounds__(LS_THREADS) void k_icp_moments(const float *__restrict__ p1, const float *__restrict__ n1,
                                                            const float *__restrict__ tgt, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ dist, const float *__restrict__ thr,
                                                            int N, int M, double *__restrict__ partial) {
  __shared__ double sh[LS_THREADS / 64];
  const int b = blockIdx.y;
  const int i = blockIdx.x * LS_THREADS + threadIdx.x;
  double mo[ICP_NM];
#pragma unroll
  for (int k = 0; k < ICP_NM; ++k) mo[k] = 0.0;
  if (i < N && dist[(int64_t)b * N + i] < thr[b]) {
    const int j = idx[(int64_t)b * N + i];
    const float *s = p1 + ((int64_t)b * N + i) * 3, *t = tgt + ((int64_t)b * M + j) * 3;
    const float *n = n1 + ((int64_t)b * N + i) * 3;
    const float e0 = t[0] - s[0], e1 = t[1] - s[1], e2 = t[2] - s[2];
    const float dot = n[0] * e0 + n[1] * e1 + n[2] * e2;
    const float nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2], ee = e0 * e0 + e1 * e1 + e2 * e2;
    // torch.nn.functional.cosine_similarity: x.y / (max(|x|, eps) * max(|y|, eps)), eps = 1e-8
    const float c = dot / (fmaxf(sqrtf(nn), 1e-8f) * fmaxf(sqrtf(ee), 1e-8f));
    const double w = (double)(fabsf(c) * fabsf(c));
    mo[0] = 1.0;
    mo[7] = w;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mo[1 + a] = s[a];
      mo[4 + a] = t[a];
      mo[8 + a] = w * s[a];
      mo[11 + a] = w * t[a];
#pragma unroll
      for (int c2 = 0; c2 < 3; ++c2) mo[14 + a * 3 + c2] = w * (double)s[a] * (double)t[c2];
    }
  }
  double *o = partial + ((int64_t)b * gridDim.x + blockIdx.x) * ICP_NM;
#pragma unroll
  for (int k = 0; k < ICP_NM; ++k) {
    const double sres = block_sum(mo[k], sh);
    if (threadIdx.x == 0) o[k] = sres;
  }
}

__device__ void svd3_jacobi(const double H[9], double U[9], double S[3], double V[9]) {
  double A[9];
  for (int k = 0; k < 9; ++k) { A[k] = H[k]; V[k] = (k % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 3; ++k) { al += A[k * 3 + p] * A[k * 3 + p]; be += A[k * 3 + q] * A[k * 3 + q]; ga += A[k * 3 + p] * A[k * 3 + q]; }
        off = fmax(off, fabs(ga) / (sqrt(al * be) + 1e-300));
        if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; ++k) {
          const double ap = A[k * 3 + p], aq = A[k * 3 + q];
          A[k * 3 + p] = c * ap - s * aq;
          A[k * 3 + q] = s * ap + c * aq;
          const double vp = V[k * 3 + p], vq = V[k * 3 + q];
          V[k * 3 + p] = c * vp - s * vq;
          V[k * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int k = 0; k < 3; ++k) S[k] = sqrt(A[0 * 3 + k] * A[0 * 3 + k] + A[1 * 3 + k] * A[1 * 3 + k] + A[2 * 3 + k] * A[2 * 3 + k]);
  // sort descending (columns of A and V)
  for (int a = 0; a < 2; ++a)
    for (int b2 = a + 1; b2 < 3; ++b2)
      if (S[b2] > S[a]) {
        const double ts = S[a]; S[a] = S[b2]; S[b2] = ts;
        for (int k = 0; k < 3; ++k) {
          double tmp = A[k * 3 + a]; A[k * 3 + a] = A[k * 3 + b2]; A[k * 3 + b2] = tmp;
          tmp = V[k * 3 + a]; V[k * 3 + a] = V[k * 3 + b2]; V[k * 3 + b2] = tmp;
        }
      }
  const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
  for (int k = 0; k < 3; ++k) {
    if (S[k] > tiny) {
      for (int r = 0; r < 3; ++r) U[r * 3 + k] = A[r * 3 + k] / S[k];
    } else if (k == 2) {   // complete to a right-handed frame; the reflection fix below decides the sign
      U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
      U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
      U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    } else {
      for (int r = 0; r < 3; ++r) U[r * 3 + k] = (r == k) ? 1.0 : 0.0;
    }
  }
}

__device__ double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// One thread per pair: reduce partials, Kabsch, compose res_r/res_t in place.
__global__ void k_icp_solve(const double *__restrict__ partial, int nblk, float *__restrict__ res_r,
                            float *__restrict__ res_t, float *__restrict__ step_R, float *__restrict__ step_t,
                            int first) {
  const int b = blockIdx.x;
  __shared__ double mo[ICP_NM];
  // one wave per moment (4 waves take the moments round-robin), lanes stride over the block partials: a single thread
  // per moment walked ~120 dependent loads (most of this kernel's 55 us); fixed lane / shuffle order: deterministic
  for (int m = threadIdx.x >> 6; m < ICP_NM; m += blockDim.x >> 6) {
    double s = 0.0;
    for (int k = threadIdx.x & 63; k < nblk; k += 64) s += partial[((int64_t)b * nblk + k) * ICP_NM + m];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) mo[m] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double cnt = mo[0] > 1.0 ? mo[0] : 1.0;
  double cs[3], ct[3];
  for (int a = 0; a < 3; ++a) { cs[a] = mo[1 + a] / cnt; ct[a] = mo[4 + a] / cnt; }
  double H[9];   // sum mw (s - cs)(t - ct)^T
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c)
      H[a * 3 + c] = mo[14 + a * 3 + c] - mo[8 + a] * ct[c] - cs[a] * mo[11 + c] + mo[7] * cs[a] * ct[c];
  double U[9], S[3], V[9];
  svd3_jacobi(H, U, S, V);
  double VUt[9];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) VUt[a * 3 + c] = V[a * 3 + 0] * U[c * 3 + 0] + V[a * 3 + 1] * U[c * 3 + 1] + V[a * 3 + 2] * U[c * 3 + 2];
  double R[9];
  if (det3(VUt) < 0.0) {
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) R[a * 3 + c] = V[a * 3 + 0] * U[c * 3 + 0] + V[a * 3 + 1] * U[c * 3 + 1] - V[a * 3 + 2] * U[c * 3 + 2];
  } else {
    for (int k = 0; k < 9; ++k) R[k] = VUt[k];
  }
  // t = -R cs + ct ; return the inverse motion (R^T, -R^T t)
  double t[3], Ri[9], ti[3];
  for (int a = 0; a < 3; ++a) t[a] = -(R[a * 3 + 0] * cs[0] + R[a * 3 + 1] * cs[1] + R[a * 3 + 2] * cs[2]) + ct[a];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) Ri[a * 3 + c] = R[c * 3 + a];
  for (int a = 0; a < 3; ++a) ti[a] = -(Ri[a * 3 + 0] * t[0] + Ri[a * 3 + 1] * t[1] + Ri[a * 3 + 2] * t[2]);
  if (step_R)
    for (int k = 0; k < 9; ++k) step_R[b * 9 + k] = (float)Ri[k];
  if (step_t)
    for (int a = 0; a < 3; ++a) step_t[b * 3 + a] = (float)ti[a];
  // res_r = Ri res_r ; res_t = Ri res_t + ti
  // first: the running motion is the identity and res_r / res_t are written without being read (same bits as a
  // caller-side eye / zeros fill: the products with 0 and 1 are exact)
  double rr[9], rt[3];
  for (int k = 0; k < 9; ++k) rr[k] = first ? (k % 4 == 0 ? 1.0 : 0.0) : (double)res_r[b * 9 + k];
  for (int a = 0; a < 3; ++a) rt[a] = first ? 0.0 : (double)res_t[b * 3 + a];
  for (int a = 0; a < 3; ++a) {
    for (int c = 0; c < 3; ++c)
      res_r[b * 9 + a * 3 + c] = (float)(Ri[a * 3 + 0] * rr[0 * 3 + c] + Ri[a * 3 + 1] * rr[1 * 3 + c] + Ri[a * 3 + 2] * rr[2 * 3 + c]);
    res_t[b * 3 + a] = (float)(Ri[a * 3 + 0] * rt[0] + Ri[a * 3 + 1] * rt[1] + Ri[a * 3 + 2] * rt[2] + ti[a]);
  }
}

extern "C" size_t rslo_icp_ws_bytes(int B, int N) {
  return (size_t)(B > 0 ? B : 1) * (size_t)rslo_cdiv(N > 0 ? N : 1, LS_THREADS) * ICP_NM * sizeof(double);
}

static int icp_step_launch(const float *p1, const float *n1, const float *tgt, const int32_t *idx, const float *dist,
                           const float *thr, int B, int N, int M, void *ws, size_t ws_bytes, float *res_r, float *res_t,
                           float *step_R, float *step_t, int first, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return RSLO_OK;
  RSLO_CHECK_ARG(N >= 1 && M >= 1, "icp_step: empty clouds");
  if (ws_bytes < rslo_icp_ws_bytes(B, N)) {
    rslo_set_error("icp_step: workspace too small");
    return RSLO_EWS;
  }
  const int nblk = (int)rslo_cdiv(N, LS_THREADS);
  hipLaunchKernelGGL(k_icp_moments, dim3(nblk, B), dim3(LS_THREADS), 0, st, p1, n1, tgt, idx, dist, thr, N, M,
                     (double *)ws);
  hipLaunchKernelGGL(k_icp_solve, dim3(B), dim3(256), 0, st, (const double *)ws, nblk, res_r, res_t, step_R, step_t,
                     first);
  RSLO_CHECK_LAUNCH("icp_step");
  return RSLO_OK;
}

extern "C" int rslo_icp_step(const float *p1, const float *n1, const float *tgt, const int32_t *idx,
                                      const float *dist, const float *thr, int B, int N, int M, void *ws,
                                      size_t ws_bytes, float *res_r /*[B,9] in/out*/, float *res_t /*[B,3] in/out*/,
                                      float *step_R /*[B,9] or NULL*/, float *step_t /*[B,3] or NULL*/, void *stream) {
  return icp_step_launch(p1, n1, tgt, idx, dist, thr, B, N, M, ws, ws_bytes, res_r, res_t, step_R, step_t, 0, stream);
}

extern "C" int rslo_icp_step_first(const float *p1, const float *n1, const float *tgt, const int32_t *idx,
                                   const float *dist, const float *thr, int B, int N, int M, void *ws, size_t ws_bytes,
                                   float *res_r /*[B,9] out*/, float *res_t /*[B,3] out*/, float *step_R, float *step_t,
                                   void *stream) {
  return icp_step_launch(p1, n1, tgt, idx, dist, thr, B, N, M, ws, ws_bytes, res_r, res_t, step_R, step_t, 1, stream);
}

// out[b][j] = R[b] x[b][j] + t[b]
__global__ void k_transform(const float *__restrict__ x, const float *__restrict__ R, const float *__restrict__ t, int M,
                            float *__restrict__ out) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const float *p = x + ((int64_t)b * M + j) * 3;
  const float *r = R + b * 9;
  float *o = out + ((int64_t)b * M + j) * 3;
#pragma unroll
  for (int a = 0; a < 3; ++a) o[a] = r[a * 3 + 0] * p[0] + r[a * 3 + 1] * p[1] + r[a * 3 + 2] * p[2] + t[b * 3 + a];
}

extern "C" int rslo_transform_points(const float *x, const float *R, const float *t, int B, int M, float *out,
                                              void *stream) {
  if (B == 0 || M == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_transform, dim3((unsigned)rslo_cdiv(M, 256), B), dim3(256), 0, (hipStream_t)stream, x, R, t, M,
                     out);
  RSLO_CHECK_LAUNCH("transform_points");
  return RSLO_OK;
}

// One histogram increment per lane with the wave's most common bins aggregated: the bin of the first active lane is
// peeled twice (one LDS atomic carrying the number of lanes that share it), the rest add one each.  The step's squared
// distances share their leading bytes, so in the first passes all 64 lanes of a wave hit ONE bin and 64 same-address
// atomics serialise (61 us per launch on values of one exponent against 38 us on spread ones, scripts/bench_roi.py).
__device__ __forceinline__ void roi_hist_add(unsigned *hist, bool on, unsigned bin) {
  unsigned long long act = __ballot(on);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (act == 0ull) break;                                   // wave-uniform
    const int lead = __ffsll((long long)act) - 1;
    const unsigned b0 = (unsigned)__shfl((int)bin, lead);
    const bool same = on && bin == b0;
    const unsigned long long m = __ballot(same);
    if ((int)(threadIdx.x & 63u) == lead) atomicAdd(&hist[b0], (unsigned)__popcll(m));
    on = on && !same;
    act &= ~m;
  }
  if (on) atomicAdd(&hist[bin], 1u);
}

// ROI threshold of the consistency loss (rslo/core/losses.py:326-334): thr[b] = max(k-th smallest of dist[b][0..cnt_b),
// 1.0) with k = 1 + int(cnt_b * ratio) (clamped to [1, cnt_b]).  Exact selection by a 4-pass MSB radix select on the
// float bit patterns (distances are >= 0, so the unsigned order of the bits is the order of the values; NaN sorts last
// like torch.sort) -- one block per pair instead of a full sort of 31 k values per pair.
__global__ __launch_bounds__(1024) void k_roi_threshold(const float *__restrict__ dist, int N,
                                                        const int32_t *__restrict__ counts, double ratio,
                                                        float *__restrict__ thr) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_k;
  const int b = blockIdx.x;
  const int cnt = counts ? counts[b] : N;
  const unsigned *d = reinterpret_cast<const unsigned *>(dist) + (int64_t)b * N;
  if (cnt <= 0) {
    if (threadIdx.x == 0) thr[b] = 1.0f;
    return;
  }
  long long k = 1 + (long long)((double)cnt * ratio);
  k = k > cnt ? cnt : (k < 1 ? 1 : k);
  if (threadIdx.x == 0) {
    s_prefix = 0u;
    s_k = (unsigned)k;
  }
  // the pair's values are read ONCE into registers (ROI_RV per thread, all loads in flight together; the index is clamped
  // instead of the load being conditional, so no branch sits around a load) and the four passes run on the registers:
  // re-reading them in every pass was a chain of cnt / 1024 dependent loads per pass, 62 us per launch for 31 k values.
  // Values past ROI_RV * 1024 (not reached by the step's clouds) are still read from memory in every pass.
  constexpr int ROI_RV = 40;
  unsigned vals[ROI_RV];
#pragma unroll
  for (int j = 0; j < ROI_RV; ++j) {
    const int i = (int)threadIdx.x + j * 1024;
    vals[j] = d[i < cnt ? i : cnt - 1];
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
    for (int j = 0; j < ROI_RV; ++j) {
      const unsigned v = vals[j];
      if (j * 1024 < cnt)                                     // block-uniform: whole rounds past the count are skipped
        roi_hist_add(hist, (int)threadIdx.x + j * 1024 < cnt && (v & himask) == prefix, (v >> shift) & 0xffu);
    }
    for (int i0 = ROI_RV * 1024; i0 < cnt; i0 += 1024) {       // block-uniform trip count (ballots inside)
      const int i = i0 + (int)threadIdx.x;
      const unsigned v = d[i < cnt ? i : cnt - 1];
      roi_hist_add(hist, i < cnt && (v & himask) == prefix, (v >> shift) & 0xffu);
    }
    __syncthreads();
    // the digit whose cumulative count first reaches k: wave 0, four bins per lane, inclusive scan over the lanes (was one
    // thread walking up to 256 dependent LDS reads per pass)
    if (threadIdx.x < 64) {
      const int l = (int)threadIdx.x;
      const unsigned kk = s_k;
      unsigned h[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = hist[4 * l + j];
      const unsigned sum = h[0] + h[1] + h[2] + h[3];
      unsigned inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)inc, o);
        if (l >= o) inc += t;
      }
      const unsigned exc = inc - sum;
      const bool mine = exc < kk && kk <= inc;
      const unsigned long long any = __ballot(mine);
      if (mine) {
        unsigned acc = exc;
        int j = 0;
        while (j < 3 && acc + h[j] < kk) acc += h[j++];
        s_k = kk - acc;
        s_prefix = prefix | ((unsigned)(4 * l + j) << shift);
      } else if (any == 0ull && l == 63) {                      // k above the total (not reachable with k <= cnt)
        s_k = kk - inc;
        s_prefix = prefix | (255u << shift);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = __uint_as_float(s_prefix);
    thr[b] = v > 1.0f ? v : 1.0f;      // torch.max(m, 1): NaN m stays NaN there; distances are never NaN on this path
  }
}

extern "C" int rslo_roi_threshold(const float *dist, int B, int N, const int32_t *counts, double ratio, float *thr,
                                  void *stream) {
  RSLO_CHECK_ARG(dist && thr && B >= 0 && N >= 1, "rslo_roi_threshold: bad arguments");
  if (B == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_roi_threshold, dim3(B), dim3(1024), 0, (hipStream_t)stream, dist, N, counts, ratio, thr);
  RSLO_CHECK_LAUNCH("k_roi_threshold");
  return RSLO_OK;
}

// out[b][j] = R[b] x[b][j] (+ t[b]) for rows of `row_stride` floats (the xyz / normal columns of the padded loss batch
// are used in place), and its backward w.r.t. the pose: dR[b] = sum_j g[b][j] x[b][j]^T, dt[b] = sum_j g[b][j]
// (block partial sums in double, last block adds them in block order: deterministic; replaces 6 tiny-tile rocBLAS
// bmm launches of 0.1-0.2 ms each).  voxel_odom_net.py:668-676 (p2_moved / n2_moved).
__global__ void k_transform_rows(const float *__restrict__ x, int row_stride, const float *__restrict__ R,
                                 const float *__restrict__ t, int M, float *__restrict__ out) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const float *p = x + ((int64_t)b * M + j) * row_stride;
  const float *r = R + b * 9;
  float *o = out + ((int64_t)b * M + j) * 3;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    o[a] = r[a * 3 + 0] * p[0] + r[a * 3 + 1] * p[1] + r[a * 3 + 2] * p[2] + (t ? t[b * 3 + a] : 0.f);
}

#define TR_THREADS 256
#define TR_MAXBLK 32          // blocks per pair: each thread walks rows j, j + blocks * 256, ... (fp64 sums)
// Two launches, no fence: a last-block finish behind __threadfence() writes back the XCD's L2, which right after the
// residual kernels is full of dirty lines (78 us for this 3 MB reduction inside the step).
__global__ __launch_bounds__(TR_THREADS) void k_transform_rows_bwd(const float *__restrict__ x, int row_stride,
                                                                   const float *__restrict__ g, int M,
                                                                   double *__restrict__ part) {
  const int b = blockIdx.y;
  double s[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) s[k] = 0.0;
  for (int j = blockIdx.x * TR_THREADS + threadIdx.x; j < M; j += gridDim.x * TR_THREADS) {
    const float *p = x + ((int64_t)b * M + j) * row_stride;
    const float *q = g + ((int64_t)b * M + j) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int c = 0; c < 3; ++c) s[a * 3 + c] += (double)q[a] * (double)p[c];
      s[9 + a] += q[a];
    }
  }
  __shared__ double red[12][TR_THREADS / 64];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    double v = 0.0;
    for (int w = 0; w < TR_THREADS / 64; ++w) v += red[threadIdx.x][w];
    part[((int64_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = v;
  }
}

__global__ void k_transform_rows_bwd_finish(const double *__restrict__ part, int nblk, float *__restrict__ dR,
                                            float *__restrict__ dt) {
  // (the partials come in through LDS with all loads in flight; the sums keep their block order: deterministic)
  __shared__ double sh[TR_MAXBLK * 12];
  const int b = blockIdx.x, k = threadIdx.x;
  for (int e = threadIdx.x; e < nblk * 12; e += blockDim.x) sh[e] = part[(int64_t)b * nblk * 12 + e];
  __syncthreads();
  if (k >= 12) return;
  double v = 0.0;
  for (int i = 0; i < nblk; ++i) v += sh[i * 12 + k];
  if (k < 9)
    dR[b * 9 + k] = (float)v;
  else
    dt[b * 3 + k - 9] = (float)v;
}

extern "C" int rslo_transform_rows(const float *x, int row_stride, const float *R, const float *t, int B, int M,
                                   float *out, void *stream) {
  RSLO_CHECK_ARG(x && R && out && row_stride >= 3, "rslo_transform_rows: bad arguments");
  if (B == 0 || M == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_transform_rows, dim3((unsigned)rslo_cdiv(M, 256), B), dim3(256), 0, (hipStream_t)stream, x,
                     row_stride, R, t, M, out);
  RSLO_CHECK_LAUNCH("transform_rows");
  return RSLO_OK;
}

static int tr_blocks(int M) {
  const int64_t n = rslo_cdiv(M > 0 ? M : 1, TR_THREADS * 4);
  return (int)(n < 1 ? 1 : (n > TR_MAXBLK ? TR_MAXBLK : n));
}

extern "C" size_t rslo_transform_rows_bwd_ws_bytes(int B, int M) {
  return (size_t)(B > 0 ? B : 1) * (size_t)tr_blocks(M) * 12 * sizeof(double);
}

extern "C" int rslo_transform_rows_bwd(const float *x, int row_stride, const float *gout, int B, int M, void *ws,
                                       size_t ws_bytes, int32_t *done, float *dR, float *dt, void *stream) {
  (void)done;       // kept in the signature (ABI): the fence-based single-launch finish it served is gone
  RSLO_CHECK_ARG(x && gout && dR && dt && row_stride >= 3 && M >= 1, "rslo_transform_rows_bwd: bad arguments");
  if (B == 0) return RSLO_OK;
  if (ws_bytes < rslo_transform_rows_bwd_ws_bytes(B, M)) {
    rslo_set_error("transform_rows_bwd: workspace too small");
    return RSLO_EWS;
  }
  const int nblk = tr_blocks(M);
  hipLaunchKernelGGL(k_transform_rows_bwd, dim3((unsigned)nblk, B), dim3(TR_THREADS), 0, (hipStream_t)stream, x,
                     row_stride, gout, M, (double *)ws);
  hipLaunchKernelGGL(k_transform_rows_bwd_finish, dim3(B), dim3(64), 0, (hipStream_t)stream, (const double *)ws, nblk, dR,
                     dt);
  RSLO_CHECK_LAUNCH("transform_rows_bwd");
  return RSLO_OK;
}

// ----------------------------------------------------------------------------------------------------
// Ragged batch assembly for the consistency loss (voxel_odom_net.py:629-651: every sample's frames are cut to the
// shortest one; here all samples of a frame become ONE zero-padded [B, Lmax, C] batch).  Rows of sample b are
// src[off[b] .. off[b] + len[b]); the backward is the inverse copy (no atomics: the ranges are disjoint).
// ----------------------------------------------------------------------------------------------------
__global__ void k_pad_rows_fwd(const float *__restrict__ src, int C, const int32_t *__restrict__ off,
                               const int32_t *__restrict__ len, int Lmax, float *__restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int64_t row = i / C;
  const int r = (int)(row % Lmax), b = (int)(row / Lmax);
  out[i] = r < len[b] ? src[((int64_t)off[b] + r) * C + c] : 0.0f;
}

__global__ void k_pad_rows_bwd(const float *__restrict__ dout, int C, const int32_t *__restrict__ off,
                               const int32_t *__restrict__ len, int B, int Lmax, float *__restrict__ dsrc,
                               int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int64_t row = i / C;
  float v = 0.0f;
  for (int b = 0; b < B; ++b) {
    const int64_t r = row - off[b];
    if (r >= 0 && r < len[b]) {
      v = dout[((int64_t)b * Lmax + r) * C + c];
      break;
    }
  }
  dsrc[i] = v;
}

// The consistency loss's operands of one frame in one launch: xyz / normal columns of the voxel features and the
// covariance head's rows, each as its own zero-padded [B,Lmax,.] block (voxel_odom_net.py:630-660 slices, concatenates and
// pads them with a dozen tensor ops).  One thread per output row.
__global__ void k_pair_rows_fwd(const float *__restrict__ feats, int F, int ncol, const float *__restrict__ conf, int Cc,
                                const int32_t *__restrict__ off, const int32_t *__restrict__ len, int Lmax,
                                float *__restrict__ xyz, float *__restrict__ nrm, float *__restrict__ cov, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const int r = (int)(row % Lmax), b = (int)(row / Lmax);
  const bool live = r < len[b];
  const int64_t s = (int64_t)off[b] + r;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    xyz[row * 3 + c] = live ? feats[s * F + c] : 0.0f;
    nrm[row * 3 + c] = live ? feats[s * F + ncol + c] : 0.0f;
  }
  for (int c = 0; c < Cc; ++c) cov[row * Cc + c] = live ? conf[s * Cc + c] : 0.0f;
}

extern "C" int rslo_pair_rows_fwd(const float *feats, int64_t N, int F, const float *conf, int Cc, const int32_t *off,
                                  const int32_t *len, int B, int Lmax, float *xyz, float *nrm, float *cov, void *stream) {
  RSLO_CHECK_ARG(feats && conf && off && len && xyz && nrm && cov && F >= 6 && Cc > 0 && B > 0 && Lmax >= 0 && N >= 0,
                 "rslo_pair_rows_fwd: bad arguments");
  const int64_t rows = (int64_t)B * Lmax;
  if (rows == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_pair_rows_fwd, dim3((unsigned)rslo_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, feats, F,
                     F > 6 ? 4 : 3, conf, Cc, off, len, Lmax, xyz, nrm, cov, rows);
  RSLO_CHECK_LAUNCH("k_pair_rows_fwd");
  return RSLO_OK;
}

extern "C" int rslo_pad_rows_fwd(const float *src, int64_t N, int C, const int32_t *off, const int32_t *len, int B,
                                 int Lmax, float *out, void *stream) {
  RSLO_CHECK_ARG(src && off && len && out && C > 0 && B > 0 && Lmax >= 0 && N >= 0, "rslo_pad_rows_fwd: bad arguments");
  const int64_t total = (int64_t)B * Lmax * C;
  if (total == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_pad_rows_fwd, dim3((unsigned)rslo_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, C,
                     off, len, Lmax, out, total);
  RSLO_CHECK_LAUNCH("k_pad_rows_fwd");
  return RSLO_OK;
}

extern "C" int rslo_pad_rows_bwd(const float *dout, int64_t N, int C, const int32_t *off, const int32_t *len, int B,
                                 int Lmax, float *dsrc, void *stream) {
  RSLO_CHECK_ARG(dout && off && len && dsrc && C > 0 && B > 0 && Lmax >= 0 && N >= 0, "rslo_pad_rows_bwd: bad arguments");
  const int64_t total = N * C;
  if (total == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_pad_rows_bwd, dim3((unsigned)rslo_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, C,
                     off, len, B, Lmax, dsrc, total);
  RSLO_CHECK_LAUNCH("k_pad_rows_bwd");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// Loss assembly (a21): everything between the per-pair / per-level partial losses and the scalar the step
// differentiates -- AdaptiveWeightedL2Loss of the voted pose against the ICP pseudo-targets (rslo/core/losses.py:144-197),
// the same reduction of the pyramid levels (voxel_odom_net.py:743-798), the consistency loss's reduce (losses.py:496-506)
// and the weighted total (voxel_odom_net.py:324-376).  All operands are a few dozen floats; issued as torch ops this was
// ~50 launches forward and ~70 backward per step.  One thread does it, in the torch formulation's operation order.
//   pose:     l_b = sum_i d_bi^2 / (n + 1e-12), n = 3 | 4;   L = w (exp(-a) sum_b l_b / (B + 1e-12) + a)
//   pyramid:  P   = sum_l cw_l (w_pT (exp(-a_pT) sum_b l_lb0 / (B + 1e-12) + a_pT) + w_pR (exp(-a_pR) ... + a_pR))
//   consist.: C   = c_scale (exp(-a_C) mean_b lb_b + a_C)            (focal_gamma = 0 everywhere: the shipped configuration)
// out[0] = T + R + P + C, out[1..4] = T, R, P, C.
// ---------------------------------------------------------------------------------------
// The operands (a few dozen floats) come in through LDS, all loads in flight together, and thread 0 then walks them in the
// torch formulation's order: read straight from global memory the walk was a chain of ~60 dependent loads -- 28 us for the
// backward kernel, in the stretch of the step where nothing else runs.  Larger batches than LT_MAXB read global memory.
#define LT_MAXB 64
struct LtStage {
  float tp[LT_MAXB * 3], tt[LT_MAXB * 3], qp[LT_MAXB * 4], qt[LT_MAXB * 4];
  float pyr[RSLO_LOSS_TAIL_MAX_LEVELS * LT_MAXB * 2], pair[LT_MAXB];
};

__device__ __forceinline__ void lt_stage(RsloLossTail &p, LtStage &S) {
  if (p.B > LT_MAXB || p.n_pairs > LT_MAXB) return;      // block-uniform
  for (int i = threadIdx.x; i < p.B * 3; i += blockDim.x) { S.tp[i] = p.t_pred[i]; S.tt[i] = p.t_tgt[i]; }
  for (int i = threadIdx.x; i < p.B * 4; i += blockDim.x) { S.qp[i] = p.q_pred[i]; S.qt[i] = p.q_tgt[i]; }
  for (int i = threadIdx.x; i < p.L * p.B * 2; i += blockDim.x) S.pyr[i] = p.pyr_loss_b[i];
  for (int i = threadIdx.x; i < p.n_pairs; i += blockDim.x) S.pair[i] = p.pair_loss[i];
  __syncthreads();
  p.t_pred = S.tp; p.t_tgt = S.tt; p.q_pred = S.qp; p.q_tgt = S.qt;
  if (p.L > 0) p.pyr_loss_b = S.pyr;
  if (p.n_pairs > 0) p.pair_loss = S.pair;
}

__global__ void k_loss_tail_fwd(RsloLossTail p, float *__restrict__ out) {
  __shared__ LtStage S;
  lt_stage(p, S);
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float inv_b = 1.0f / ((float)p.B + 1e-12f);
  float st = 0.f, sr = 0.f;
  for (int b = 0; b < p.B; ++b) {
    float a = 0.f, c = 0.f;
    for (int i = 0; i < 3; ++i) {
      const float d = p.t_pred[b * 3 + i] - p.t_tgt[b * 3 + i];
      a += d * d;
    }
    for (int i = 0; i < 4; ++i) {
      const float d = p.q_pred[b * 4 + i] - p.q_tgt[b * 4 + i];
      c += d * d;
    }
    st += a / (3.0f + 1e-12f);
    sr += c / (4.0f + 1e-12f);
  }
  const float aT = *p.alpha_T, aR = *p.alpha_R;
  const float T = p.w_T * (expf(-aT) * st * inv_b + aT), R = p.w_R * (expf(-aR) * sr * inv_b + aR);
  float P = 0.f;
  if (p.L > 0) {
    const float apT = *p.alpha_pT, apR = *p.alpha_pR;
    for (int l = 0; l < p.L; ++l) {
      float s0 = 0.f, s1 = 0.f;
      for (int b = 0; b < p.B; ++b) {
        s0 += p.pyr_loss_b[(l * p.B + b) * 2 + 0];
        s1 += p.pyr_loss_b[(l * p.B + b) * 2 + 1];
      }
      P += p.level_w[l] * (p.w_pT * (expf(-apT) * s0 * inv_b + apT) + p.w_pR * (expf(-apR) * s1 * inv_b + apR));
    }
  }
  float Cl = 0.f;
  if (p.n_pairs > 0) {
    float s = 0.f;
    for (int b = 0; b < p.n_pairs; ++b) s += p.pair_loss[b];
    const float aC = *p.alpha_C;
    Cl = p.c_scale * (expf(-aC) * (s / (float)p.n_pairs) + aC);
  }
  out[0] = T + R + P + Cl;
  out[1] = T;
  out[2] = R;
  out[3] = P;
  out[4] = Cl;
}

// gradients of out[0] scaled by *g: d_t [B,3], d_q [B,4], d_pyr [L,B,2], d_pair [n_pairs], d_alpha = (T, R, pT, pR, C), entry i at d_alpha[4 i]
// A module used for more than one term (the same alpha pointer twice): the FIRST of its entries holds the sum of all of
// them, added in entry order -- what the autograd engine's accumulation of the separate entries gives, without its launches.
__global__ void k_loss_tail_bwd(RsloLossTail p, const float *__restrict__ g, float *__restrict__ d_t,
                                float *__restrict__ d_q, float *__restrict__ d_pyr, float *__restrict__ d_pair,
                                float *__restrict__ d_alpha) {
  __shared__ LtStage S;
  const float go = *g, inv_b = 1.0f / ((float)p.B + 1e-12f);
  const float aT = *p.alpha_T, aR = *p.alpha_R;
  const float eT = expf(-aT), eR = expf(-aR);
  const float *const alpha_ptr[5] = {p.alpha_T, p.alpha_R, p.alpha_pT, p.alpha_pR, p.alpha_C};
  lt_stage(p, S);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  const float kt = go * p.w_T * eT * inv_b * 2.0f / (3.0f + 1e-12f), kq = go * p.w_R * eR * inv_b * 2.0f / (4.0f + 1e-12f);
  for (int i = tid; i < p.B * 3; i += nth) d_t[i] = kt * (p.t_pred[i] - p.t_tgt[i]);
  for (int i = tid; i < p.B * 4; i += nth) d_q[i] = kq * (p.q_pred[i] - p.q_tgt[i]);
  if (p.L > 0) {
    const float epT = expf(-*p.alpha_pT), epR = expf(-*p.alpha_pR);
    for (int i = tid; i < p.L * p.B * 2; i += nth) {
      const int l = i / (p.B * 2);
      d_pyr[i] = go * p.level_w[l] * ((i & 1) ? p.w_pR * epR : p.w_pT * epT) * inv_b;
    }
  }
  if (p.n_pairs > 0) {
    const float kc = go * p.c_scale * expf(-*p.alpha_C) / (float)p.n_pairs;
    for (int i = tid; i < p.n_pairs; i += nth) d_pair[i] = kc;
  }
  if (tid == 0) {
    float st = 0.f, sr = 0.f;
    for (int b = 0; b < p.B; ++b) {
      float a = 0.f, c = 0.f;
      for (int i = 0; i < 3; ++i) {
        const float d = p.t_pred[b * 3 + i] - p.t_tgt[b * 3 + i];
        a += d * d;
      }
      for (int i = 0; i < 4; ++i) {
        const float d = p.q_pred[b * 4 + i] - p.q_tgt[b * 4 + i];
        c += d * d;
      }
      st += a / (3.0f + 1e-12f);
      sr += c / (4.0f + 1e-12f);
    }
    const float dT = go * p.w_T * (1.0f - eT * st * inv_b);
    const float dR = go * p.w_R * (1.0f - eR * sr * inv_b);
    float dpT = 0.f, dpR = 0.f;
    if (p.L > 0) {
      const float epT = expf(-*p.alpha_pT), epR = expf(-*p.alpha_pR);
      for (int l = 0; l < p.L; ++l) {
        float s0 = 0.f, s1 = 0.f;
        for (int b = 0; b < p.B; ++b) {
          s0 += p.pyr_loss_b[(l * p.B + b) * 2 + 0];
          s1 += p.pyr_loss_b[(l * p.B + b) * 2 + 1];
        }
        dpT += p.level_w[l] * p.w_pT * (1.0f - epT * s0 * inv_b);
        dpR += p.level_w[l] * p.w_pR * (1.0f - epR * s1 * inv_b);
      }
    }
    float dc = 0.f;
    if (p.n_pairs > 0) {
      float s = 0.f;
      for (int b = 0; b < p.n_pairs; ++b) s += p.pair_loss[b];
      dc = go * p.c_scale * (1.0f - expf(-*p.alpha_C) * (s / (float)p.n_pairs));
    }
    float da[5] = {dT, dR, go * dpT, go * dpR, dc};
    for (int i = 1; i < 5; ++i)
      for (int j = 0; j < i; ++j)
        if (alpha_ptr[i] && alpha_ptr[i] == alpha_ptr[j]) {
          da[j] += da[i];
          break;
        }
    for (int i = 0; i < 5; ++i) d_alpha[i * RSLO_LOSS_TAIL_ALPHA_STRIDE] = da[i];
  }
}

static int loss_tail_check(const RsloLossTail *p) {
  RSLO_CHECK_ARG(p && p->B >= 1 && p->B <= 4096 && p->L >= 0 && p->L <= RSLO_LOSS_TAIL_MAX_LEVELS && p->n_pairs >= 0,
                 "rslo_loss_tail: bad sizes");
  RSLO_CHECK_ARG(p->t_pred && p->t_tgt && p->q_pred && p->q_tgt && p->alpha_T && p->alpha_R, "rslo_loss_tail: null pose operand");
  RSLO_CHECK_ARG(p->L == 0 || (p->pyr_loss_b && p->alpha_pT && p->alpha_pR), "rslo_loss_tail: null pyramid operand");
  RSLO_CHECK_ARG(p->n_pairs == 0 || (p->pair_loss && p->alpha_C), "rslo_loss_tail: null consistency operand");
  return RSLO_OK;
}

extern "C" int rslo_loss_tail_fwd(const RsloLossTail *h_p, float *out5, void *stream) {
  if (int rc = loss_tail_check(h_p)) return rc;
  RSLO_CHECK_ARG(out5, "rslo_loss_tail_fwd: null output");
  hipLaunchKernelGGL(k_loss_tail_fwd, dim3(1), dim3(64), 0, (hipStream_t)stream, *h_p, out5);
  RSLO_CHECK_LAUNCH("k_loss_tail_fwd");
  return RSLO_OK;
}

extern "C" int rslo_loss_tail_bwd(const RsloLossTail *h_p, const float *grad_out, float *d_t, float *d_q, float *d_pyr,
                                  float *d_pair, float *d_alpha5, void *stream) {
  if (int rc = loss_tail_check(h_p)) return rc;
  RSLO_CHECK_ARG(grad_out && d_t && d_q && d_alpha5 && (h_p->L == 0 || d_pyr) && (h_p->n_pairs == 0 || d_pair),
                 "rslo_loss_tail_bwd: null output");
  hipLaunchKernelGGL(k_loss_tail_bwd, dim3(1), dim3(256), 0, (hipStream_t)stream, *h_p, grad_out, d_t, d_q, d_pyr,
                     d_pair, d_alpha5);
  RSLO_CHECK_LAUNCH("k_loss_tail_bwd");
  return RSLO_OK;
}
