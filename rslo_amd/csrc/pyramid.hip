// Pyramid (per-geometric-unit) L2 supervision of the local transformation maps, all levels in one launch
// (voxel_odom_net.py:706-760 create_loss pyramid part, gen_tq_maps voxel_odom_net.py:575-600,
//  generate_pointwise_local_transformation_tch rslo/data/dataset.py:118-168, AdaptiveWeightedL2Loss
//  rslo/core/losses.py:144-197).
//
// The reference materialises the target map [B,7,H0,W0] (the local transform every BEV cell would have to
// predict for the global pose tq_b), nearest-resamples it to every pyramid level, and runs
// sum(diff^2 * mask) / (sum(mask) + 1e-12) per sample with ~25 elementwise launches per level.  Here the target
// of a cell is recomputed in registers from tq_b and the cell index (7 floats of input per sample instead of
// a 7-channel map), so the forward reads pred + mask exactly once and the backward writes dpred exactly once:
// HBM-bound elementwise work, bytes = cells * (7 + Cm) * 4 forward, cells * (14 + Cm) * 4 backward.
//
// Deterministic: block partial sums in double, the last block of a (level, sample) row reduces them in block order.
#include "rslo_common.h"

#define PY_THREADS 256
#define PY_MAX_LEVELS 4

struct PyLevels {
  const float *pred[PY_MAX_LEVELS];
  const float *mask[PY_MAX_LEVELS];
  float *dpred[PY_MAX_LEVELS];
  int h[PY_MAX_LEVELS], w[PY_MAX_LEVELS], cm[PY_MAX_LEVELS];
  int n;
};

struct PyGeom {
  int H0, W0;
  float ox, oy, oz, vx, vy, vz;
};

// local target (t_l, q) of cell (i, j) of an h x w level: nearest source cell of the H0 x W0 map
// (torch 'nearest': src = min(floor(dst * (float)in / out), in - 1)), anchored at
// x = (sj - ox) vx, y = (oy - si) vy, z = (0 - oz) vz;  t_l = R(q)^-1 (t_g - x) + x.
__device__ __forceinline__ void py_target(const float *__restrict__ tq, PyGeom g, int h, int w, int i, int j,
                                          float out[7]) {
  const float sh = (float)g.H0 / (float)h, sw = (float)g.W0 / (float)w;
  int si = (int)floorf((float)i * sh), sj = (int)floorf((float)j * sw);
  si = si < g.H0 - 1 ? si : g.H0 - 1;
  sj = sj < g.W0 - 1 ? sj : g.W0 - 1;
  const float cx = ((float)sj - g.ox) * g.vx;
  const float cy = (-(float)si + g.oy) * g.vy;
  const float cz = (0.0f - g.oz) * g.vz;
  const float qs = tq[3], qx = -tq[4], qy = -tq[5], qz = -tq[6];   // inverse rotation
  const float tx = tq[0] - cx, ty = tq[1] - cy, tz = tq[2] - cz;
  // t' = t + 2 qs (qv x t) + 2 qv x (qv x t)   (rslo/utils/pose_utils.py:130-142)
  const float bx = qy * tz - qz * ty, by = qz * tx - qx * tz, bz = qx * ty - qy * tx;
  const float c0 = 2.0f * (qy * bz - qz * by), c1 = 2.0f * (qz * bx - qx * bz), c2 = 2.0f * (qx * by - qy * bx);
  out[0] = tx + 2.0f * bx * qs + c0 + cx;
  out[1] = ty + 2.0f * by * qs + c1 + cy;
  out[2] = tz + 2.0f * bz * qs + c2 + cz;
  out[3] = tq[3];
  out[4] = tq[4];
  out[5] = tq[5];
  out[6] = tq[6];
}

// grid: (blocks_x, n_levels * B).  part: [L*B][blocks_x][4] doubles; done: [L*B] counters (zero on entry, reset
// to zero by the last block); loss_b: [L,B,2], den: [L,B,2].
__global__ __launch_bounds__(PY_THREADS) void k_pyramid_fwd(PyLevels lv, int B, const float *__restrict__ tq,
                                                            PyGeom g, double *__restrict__ part,
                                                            int *__restrict__ done, float *__restrict__ loss_b,
                                                            float *__restrict__ den) {
  const int row = blockIdx.y, l = row / B, b = row % B;
  const int h = lv.h[l], w = lv.w[l], cm = lv.cm[l];
  const int cells = h * w;
  const int nblk = (cells + PY_THREADS - 1) / PY_THREADS;
  if ((int)blockIdx.x >= nblk) return;
  const float *__restrict__ pred = lv.pred[l] + (size_t)b * 7 * cells;
  const float *__restrict__ mask = lv.mask[l] + (size_t)b * cm * cells;
  const int c = blockIdx.x * PY_THREADS + threadIdx.x;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < cells) {
    float tgt[7];
    py_target(tq + b * 7, g, h, w, c / w, c % w, tgt);
    const float mT = mask[c], mR = mask[(size_t)(cm - 1) * cells + c];
    float aT = 0.f, aR = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float d = pred[(size_t)k * cells + c] - tgt[k];
      aT += d * d * mT;
    }
#pragma unroll
    for (int k = 3; k < 7; ++k) {
      const float d = pred[(size_t)k * cells + c] - tgt[k];
      aR += d * d * mR;
    }
    s[0] = aT;
    s[1] = aR;
    s[2] = 3.0 * (double)mT;
    s[3] = 4.0 * (double)mR;
  }
  __shared__ double red[4][PY_THREADS / 64];
  __shared__ int is_last;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0.0;
    for (int k = 0; k < PY_THREADS / 64; ++k) v += red[threadIdx.x][k];
    part[((size_t)row * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {      // one lane releases / acquires for the block (an all-lane fence costs 2-4x)
    __threadfence();
    is_last = (atomicAdd(&done[row], 1) == nblk - 1);
    if (is_last) __threadfence();
  }
  __syncthreads();
  if (!is_last) return;
  if (threadIdx.x < 4) {
    double v = 0.0;
    for (int k = 0; k < nblk; ++k) v += part[((size_t)row * gridDim.x + k) * 4 + threadIdx.x];
    red[threadIdx.x][0] = v;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float num = (float)red[threadIdx.x][0], dn = (float)red[2 + threadIdx.x][0];
    loss_b[(size_t)row * 2 + threadIdx.x] = num / (dn + 1e-12f);
    den[(size_t)row * 2 + threadIdx.x] = dn;
  }
  if (threadIdx.x == 0) done[row] = 0;
}

// dpred[b,k,c] = g[l,b,T|R] * 2 * diff * mask / (den + 1e-12)
__global__ __launch_bounds__(PY_THREADS) void k_pyramid_bwd(PyLevels lv, int B, const float *__restrict__ tq,
                                                            PyGeom g, const float *__restrict__ gl,
                                                            const float *__restrict__ den) {
  const int row = blockIdx.y, l = row / B, b = row % B;
  const int h = lv.h[l], w = lv.w[l], cm = lv.cm[l];
  const int cells = h * w;
  const int c = blockIdx.x * PY_THREADS + threadIdx.x;
  if (c >= cells) return;
  const float *__restrict__ pred = lv.pred[l] + (size_t)b * 7 * cells;
  const float *__restrict__ mask = lv.mask[l] + (size_t)b * cm * cells;
  float *__restrict__ dp = lv.dpred[l] + (size_t)b * 7 * cells;
  float tgt[7];
  py_target(tq + b * 7, g, h, w, c / w, c % w, tgt);
  const float kT = 2.0f * gl[(size_t)row * 2] / (den[(size_t)row * 2] + 1e-12f) * mask[c];
  const float kR = 2.0f * gl[(size_t)row * 2 + 1] / (den[(size_t)row * 2 + 1] + 1e-12f) *
                   mask[(size_t)(cm - 1) * cells + c];
#pragma unroll
  for (int k = 0; k < 7; ++k) dp[(size_t)k * cells + c] = (pred[(size_t)k * cells + c] - tgt[k]) * (k < 3 ? kT : kR);
}

static int py_pack(const RsloPyramidLevel *levels, int n_levels, PyLevels &lv, int &max_cells) {
  lv.n = n_levels;
  max_cells = 0;
  for (int l = 0; l < n_levels; ++l) {
    lv.pred[l] = levels[l].pred;
    lv.mask[l] = levels[l].mask;
    lv.dpred[l] = levels[l].dpred;
    lv.h[l] = levels[l].h;
    lv.w[l] = levels[l].w;
    lv.cm[l] = levels[l].mask_channels;
    if (levels[l].h <= 0 || levels[l].w <= 0 || levels[l].mask_channels <= 0 || !levels[l].pred || !levels[l].mask)
      return -1;
    const int cells = levels[l].h * levels[l].w;
    max_cells = cells > max_cells ? cells : max_cells;
  }
  return 0;
}

extern "C" size_t rslo_pyramid_l2_ws_bytes(const RsloPyramidLevel *levels, int n_levels, int B) {
  int max_cells = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int cells = levels[l].h * levels[l].w;
    max_cells = cells > max_cells ? cells : max_cells;
  }
  const size_t nblk = (size_t)rslo_cdiv(max_cells > 0 ? max_cells : 1, PY_THREADS);
  return (size_t)n_levels * B * nblk * 4 * sizeof(double);
}

extern "C" int rslo_pyramid_l2_fwd(const RsloPyramidLevel *levels, int n_levels, int B, const float *tq, int H0,
                                   int W0, const float *h_origin3, const float *h_vsize3, void *ws, size_t ws_bytes,
                                   int32_t *done, float *loss_b, float *den, void *stream) {
  RSLO_CHECK_ARG(n_levels >= 1 && n_levels <= PY_MAX_LEVELS, "rslo_pyramid_l2_fwd: 1..%d levels", PY_MAX_LEVELS);
  RSLO_CHECK_ARG(B >= 1 && tq && done && loss_b && den && H0 > 0 && W0 > 0, "rslo_pyramid_l2_fwd: bad arguments");
  PyLevels lv;
  int max_cells;
  RSLO_CHECK_ARG(py_pack(levels, n_levels, lv, max_cells) == 0, "rslo_pyramid_l2_fwd: bad level descriptor");
  RSLO_CHECK_ARG(ws && ws_bytes >= rslo_pyramid_l2_ws_bytes(levels, n_levels, B), "rslo_pyramid_l2_fwd: workspace");
  PyGeom g = {H0, W0, h_origin3[0], h_origin3[1], h_origin3[2], h_vsize3[0], h_vsize3[1], h_vsize3[2]};
  dim3 grid((unsigned)rslo_cdiv(max_cells, PY_THREADS), (unsigned)(n_levels * B));
  hipLaunchKernelGGL(k_pyramid_fwd, grid, dim3(PY_THREADS), 0, (hipStream_t)stream, lv, B, tq, g, (double *)ws,
                     (int *)done, loss_b, den);
  RSLO_CHECK_LAUNCH("k_pyramid_fwd");
  return RSLO_OK;
}

extern "C" int rslo_pyramid_l2_bwd(const RsloPyramidLevel *levels, int n_levels, int B, const float *tq, int H0,
                                   int W0, const float *h_origin3, const float *h_vsize3, const float *grad_loss_b,
                                   const float *den, void *stream) {
  RSLO_CHECK_ARG(n_levels >= 1 && n_levels <= PY_MAX_LEVELS, "rslo_pyramid_l2_bwd: 1..%d levels", PY_MAX_LEVELS);
  RSLO_CHECK_ARG(B >= 1 && tq && grad_loss_b && den, "rslo_pyramid_l2_bwd: bad arguments");
  PyLevels lv;
  int max_cells;
  RSLO_CHECK_ARG(py_pack(levels, n_levels, lv, max_cells) == 0, "rslo_pyramid_l2_bwd: bad level descriptor");
  for (int l = 0; l < n_levels; ++l) RSLO_CHECK_ARG(lv.dpred[l], "rslo_pyramid_l2_bwd: dpred missing");
  PyGeom g = {H0, W0, h_origin3[0], h_origin3[1], h_origin3[2], h_vsize3[0], h_vsize3[1], h_vsize3[2]};
  dim3 grid((unsigned)rslo_cdiv(max_cells, PY_THREADS), (unsigned)(n_levels * B));
  hipLaunchKernelGGL(k_pyramid_bwd, grid, dim3(PY_THREADS), 0, (hipStream_t)stream, lv, B, tq, g, grad_loss_b, den);
  RSLO_CHECK_LAUNCH("k_pyramid_bwd");
  return RSLO_OK;
}
