// Local -> global transformation of every BEV cell and the confidence-weighted ego-motion vote, fused
// (from_pointwise_local_transformation_tch rslo/data/dataset.py:121-208 with rotate_vec_by_q
//  rslo/utils/pose_utils.py:130-142; aggregate_tq non-SVD branch rslo/models/odom_pred.py:347-357).
//
//   x      = centre of cell (i, j):  ((j - ox) vx, (oy - i) vy, (0 - oz) vz)
//   t_g    = t_l - x rotated by q (as given, no normalisation inside the rotation) + x
//   q_g    = q / max(|q|, 1e-12)
//   t      = sum t_g c_t / (sum c_t + 1e-12),   qv = sum q_g c_r / (sum c_r + 1e-12)
//
// The reference runs ~40 elementwise / reduction launches forward and ~60 backward on [B,7,96,176] maps; here one
// kernel writes the global map and the 9 weighted sums per sample (block partial sums in double, the last block of a
// sample finishes them in block order: deterministic), and one elementwise kernel produces the gradients of the
// local map and of both confidence maps from the gradient of the voted pose.  HBM-bound: forward reads 9 and writes
// 7 floats per cell, backward reads 9 and writes 9.
#include "rslo_common.h"

#define VT_THREADS 256

struct VoteGeom {
  int H, W;
  float ox, oy, oz, vx, vy, vz;
};

struct VoteCell {
  float v[3];    // t_l - x
  float q[4];    // w, x, y, z as given
  float tg[3];
  float qg[4];
  float n;       // max(|q|, 1e-12)
};

__device__ __forceinline__ void vote_cell(const float *__restrict__ tq, int64_t cells, int64_t c, VoteGeom g, VoteCell &o) {
  const int i = (int)(c / g.W), j = (int)(c - (int64_t)i * g.W);
  const float cx = ((float)j - g.ox) * g.vx, cy = (-(float)i + g.oy) * g.vy, cz = (0.0f - g.oz) * g.vz;
  o.v[0] = tq[0 * cells + c] - cx;
  o.v[1] = tq[1 * cells + c] - cy;
  o.v[2] = tq[2 * cells + c] - cz;
#pragma unroll
  for (int k = 0; k < 4; ++k) o.q[k] = tq[(3 + k) * cells + c];
  const float qs = o.q[0], qx = o.q[1], qy = o.q[2], qz = o.q[3];
  const float bx = qy * o.v[2] - qz * o.v[1], by = qz * o.v[0] - qx * o.v[2], bz = qx * o.v[1] - qy * o.v[0];
  const float c0 = 2.0f * (qy * bz - qz * by), c1 = 2.0f * (qz * bx - qx * bz), c2 = 2.0f * (qx * by - qy * bx);
  o.tg[0] = o.v[0] + 2.0f * bx * qs + c0 + cx;
  o.tg[1] = o.v[1] + 2.0f * by * qs + c1 + cy;
  o.tg[2] = o.v[2] + 2.0f * bz * qs + c2 + cz;
  const float n = sqrtf(qs * qs + qx * qx + qy * qy + qz * qz);
  o.n = fmaxf(n, 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; ++k) o.qg[k] = o.q[k] / o.n;
}

// grid (blocks, B).  part [B][blocks][9] doubles; done [B] zero on entry / exit.
// odom [B,7]; sums [B,2] = (sum c_t + 1e-12, sum c_r + 1e-12) kept for the backward.
__global__ __launch_bounds__(VT_THREADS) void k_vote_fwd(const float *__restrict__ tq, const float *__restrict__ tconf,
                                                         const float *__restrict__ rconf, VoteGeom g,
                                                         float *__restrict__ tq_g, double *__restrict__ part,
                                                         int *__restrict__ done, float *__restrict__ odom,
                                                         float *__restrict__ sums) {
  const int b = blockIdx.y;
  const int64_t cells = (int64_t)g.H * g.W;
  const int64_t c = (int64_t)blockIdx.x * VT_THREADS + threadIdx.x;
  double s[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = 0.0;
  if (c < cells) {
    VoteCell o;
    vote_cell(tq + (int64_t)b * 7 * cells, cells, c, g, o);
    float *og = tq_g + (int64_t)b * 7 * cells;
#pragma unroll
    for (int k = 0; k < 3; ++k) og[k * cells + c] = o.tg[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) og[(3 + k) * cells + c] = o.qg[k];
    const float ct = tconf[(int64_t)b * cells + c], cr = rconf[(int64_t)b * cells + c];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = (double)(o.tg[k] * ct);
    s[3] = ct;
#pragma unroll
    for (int k = 0; k < 4; ++k) s[4 + k] = (double)(o.qg[k] * cr);
    s[8] = cr;
  }
  __shared__ double red[9][VT_THREADS / 64];
  __shared__ int is_last;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    double v = 0.0;
    for (int w = 0; w < VT_THREADS / 64; ++w) v += red[threadIdx.x][w];
    part[((int64_t)b * gridDim.x + blockIdx.x) * 9 + threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    is_last = atomicAdd(&done[b], 1) == (int)gridDim.x - 1;
    if (is_last) __threadfence();
  }
  __syncthreads();
  if (!is_last) return;
  __shared__ double tot[9];
  if (threadIdx.x < 9) {
    double v = 0.0;
    for (unsigned k = 0; k < gridDim.x; ++k) v += part[((int64_t)b * gridDim.x + k) * 9 + threadIdx.x];
    tot[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float st = (float)tot[3] + 1e-12f, sr = (float)tot[8] + 1e-12f;
    for (int k = 0; k < 3; ++k) odom[b * 7 + k] = (float)tot[k] / st;
    for (int k = 0; k < 4; ++k) odom[b * 7 + 3 + k] = (float)tot[4 + k] / sr;
    sums[b * 2 + 0] = st;
    sums[b * 2 + 1] = sr;
    done[b] = 0;
  }
}

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// gradients of the voted pose (g_odom [B,7]) w.r.t. the local map, t_conf and r_conf
__global__ __launch_bounds__(VT_THREADS) void k_vote_bwd(const float *__restrict__ tq, const float *__restrict__ tconf,
                                                         const float *__restrict__ rconf, VoteGeom g,
                                                         const float *__restrict__ odom, const float *__restrict__ sums,
                                                         const float *__restrict__ g_odom, float *__restrict__ d_tq,
                                                         float *__restrict__ d_tconf, float *__restrict__ d_rconf) {
  const int b = blockIdx.y;
  const int64_t cells = (int64_t)g.H * g.W;
  const int64_t c = (int64_t)blockIdx.x * VT_THREADS + threadIdx.x;
  if (c >= cells) return;
  VoteCell o;
  vote_cell(tq + (int64_t)b * 7 * cells, cells, c, g, o);
  const float st = sums[b * 2 + 0], sr = sums[b * 2 + 1];
  const float ct = tconf[(int64_t)b * cells + c], cr = rconf[(int64_t)b * cells + c];
  const float *go = g_odom + b * 7, *od = odom + b * 7;
  // confidences: d t / d c_t = (t_g - t) / S_t
  float dct = 0.f, dcr = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) dct += go[k] * (o.tg[k] - od[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) dcr += go[3 + k] * (o.qg[k] - od[3 + k]);
  d_tconf[(int64_t)b * cells + c] = dct / st;
  d_rconf[(int64_t)b * cells + c] = dcr / sr;
  // global translation -> local translation and quaternion (rotation part)
  const float gt[3] = {go[0] * ct / st, go[1] * ct / st, go[2] * ct / st};
  const float qs = o.q[0];
  const float qv[3] = {o.q[1], o.q[2], o.q[3]};
  float a[3], a2[3];
  cross3(qv, gt, a);            // qv x g
  cross3(qv, a, a2);            // qv x (qv x g)
  float dtl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) dtl[k] = gt[k] - 2.0f * qs * a[k] + 2.0f * a2[k];
  float bvec[3], vxg[3], gxq[3], bxg[3], vxgxq[3];
  cross3(qv, o.v, bvec);        // b = qv x v
  cross3(o.v, gt, vxg);         // v x g
  cross3(gt, qv, gxq);          // g x qv
  cross3(bvec, gt, bxg);        // b x g
  cross3(o.v, gxq, vxgxq);      // v x (g x qv)
  const float dqs = 2.0f * (bvec[0] * gt[0] + bvec[1] * gt[1] + bvec[2] * gt[2]);
  float dq[4];
  dq[0] = dqs;
#pragma unroll
  for (int k = 0; k < 3; ++k) dq[1 + k] = 2.0f * qs * vxg[k] + 2.0f * bxg[k] + 2.0f * vxgxq[k];
  // normalised quaternion part: d q = (d q_g - q_g (q_g . d q_g)) / n   (zero where |q| < 1e-12: the clamp is flat)
  float gq[4], dot = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    gq[k] = go[3 + k] * cr / sr;
    dot += o.qg[k] * gq[k];
  }
  const float n_raw = sqrtf(o.q[0] * o.q[0] + o.q[1] * o.q[1] + o.q[2] * o.q[2] + o.q[3] * o.q[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) dq[k] += (n_raw >= 1e-12f) ? (gq[k] - o.qg[k] * dot) / o.n : gq[k] / o.n;
  float *dd = d_tq + (int64_t)b * 7 * cells;
#pragma unroll
  for (int k = 0; k < 3; ++k) dd[k * cells + c] = dtl[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) dd[(3 + k) * cells + c] = dq[k];
}

extern "C" size_t rslo_vote_ws_bytes(int B, int H, int W) {
  return (size_t)(B > 0 ? B : 1) * (size_t)rslo_cdiv((int64_t)H * W, VT_THREADS) * 9 * sizeof(double);
}

static VoteGeom vote_geom(int H, int W, const float *o3, const float *v3) {
  VoteGeom g = {H, W, o3[0], o3[1], o3[2], v3[0], v3[1], v3[2]};
  return g;
}

extern "C" int rslo_vote_fwd(const float *tq_map, const float *t_conf, const float *r_conf, int B, int H, int W,
                             const float *h_origin3, const float *h_vsize3, void *ws, size_t ws_bytes, int32_t *done,
                             float *tq_map_g, float *odom, float *sums, void *stream) {
  RSLO_CHECK_ARG(tq_map && t_conf && r_conf && done && tq_map_g && odom && sums && H > 0 && W > 0,
                 "rslo_vote_fwd: bad arguments");
  if (B == 0) return RSLO_OK;
  RSLO_CHECK_ARG(ws && ws_bytes >= rslo_vote_ws_bytes(B, H, W), "rslo_vote_fwd: workspace too small");
  dim3 grid((unsigned)rslo_cdiv((int64_t)H * W, VT_THREADS), (unsigned)B);
  hipLaunchKernelGGL(k_vote_fwd, grid, dim3(VT_THREADS), 0, (hipStream_t)stream, tq_map, t_conf, r_conf,
                     vote_geom(H, W, h_origin3, h_vsize3), tq_map_g, (double *)ws, (int *)done, odom, sums);
  RSLO_CHECK_LAUNCH("k_vote_fwd");
  return RSLO_OK;
}

extern "C" int rslo_vote_bwd(const float *tq_map, const float *t_conf, const float *r_conf, int B, int H, int W,
                             const float *h_origin3, const float *h_vsize3, const float *odom, const float *sums,
                             const float *g_odom, float *d_tq_map, float *d_t_conf, float *d_r_conf, void *stream) {
  RSLO_CHECK_ARG(tq_map && t_conf && r_conf && odom && sums && g_odom && d_tq_map && d_t_conf && d_r_conf,
                 "rslo_vote_bwd: bad arguments");
  if (B == 0) return RSLO_OK;
  dim3 grid((unsigned)rslo_cdiv((int64_t)H * W, VT_THREADS), (unsigned)B);
  hipLaunchKernelGGL(k_vote_bwd, grid, dim3(VT_THREADS), 0, (hipStream_t)stream, tq_map, t_conf, r_conf,
                     vote_geom(H, W, h_origin3, h_vsize3), odom, sums, g_odom, d_tq_map, d_t_conf, d_r_conf);
  RSLO_CHECK_LAUNCH("k_vote_bwd");
  return RSLO_OK;
}
