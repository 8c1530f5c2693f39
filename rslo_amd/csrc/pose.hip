// Per-pair pose algebra of the loss assembly, one thread per pair (B is the number of frame pairs of a step, <= a few
// dozen): each of these is ~25-45 tiny elementwise launches in the reference formulation.
//  * quaternion (w,x,y,z) -> rotation matrix with kornia 0.4.0 semantics (L2-normalise with eps 1e-12 first;
//    rslo/models/voxel_odom_net.py:675 via torchplus.roll) and its analytic backward;
//  * pseudo-targets from the ICP refinement (voxel_odom_net.py:709-735):
//      R* = res_R R_pred, q* = matrix_to_quaternion(R*) rolled to (w,x,y,z), sign-fixed to w >= 0,
//      t* = res_R T_pred + res_t,
//    matrix -> quaternion with kornia's four trace branches and eps = 1e-8.
#include "rslo_common.h"

__global__ void k_quat_to_rot(const float *__restrict__ q_wxyz, int B, float *__restrict__ R) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *q = q_wxyz + b * 4;
  const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
  const float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  float *r = R + b * 9;
  r[0] = 1.f - (tyy + tzz); r[1] = txy - twz;         r[2] = txz + twy;
  r[3] = txy + twz;         r[4] = 1.f - (txx + tzz); r[5] = tyz - twx;
  r[6] = txz - twy;         r[7] = tyz + twx;         r[8] = 1.f - (txx + tyy);
}

__global__ void k_quat_to_rot_bwd(const float *__restrict__ q_wxyz, const float *__restrict__ gR, int B,
                                  float *__restrict__ gq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *q = q_wxyz + b * 4, *g = gR + b * 9;
  const float nr = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float n = fmaxf(nr, 1e-12f);
  const float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  const float dx = 2.f * (y * (g[1] + g[3]) + z * (g[2] + g[6]) - 2.f * x * (g[4] + g[8]) + w * (g[7] - g[5]));
  const float dy = 2.f * (x * (g[1] + g[3]) + z * (g[5] + g[7]) - 2.f * y * (g[0] + g[8]) + w * (g[2] - g[6]));
  const float dz = 2.f * (x * (g[2] + g[6]) + y * (g[5] + g[7]) - 2.f * z * (g[0] + g[4]) + w * (g[3] - g[1]));
  const float dw = 2.f * (x * (g[7] - g[5]) + y * (g[2] - g[6]) + z * (g[3] - g[1]));
  float *o = gq + b * 4;
  if (nr >= 1e-12f) {
    const float dot = w * dw + x * dx + y * dy + z * dz;
    o[0] = (dw - w * dot) / n;
    o[1] = (dx - x * dot) / n;
    o[2] = (dy - y * dot) / n;
    o[3] = (dz - z * dot) / n;
  } else {
    o[0] = dw / n; o[1] = dx / n; o[2] = dy / n; o[3] = dz / n;
  }
}

__device__ __forceinline__ float pose_safe_div(float num, float den) {
  return num / fmaxf(den, 1.17549435e-38f);      // torch.finfo(float32).tiny
}

__global__ void k_pose_targets(const float *__restrict__ res_r, const float *__restrict__ res_t,
                               const float *__restrict__ R_pred, const float *__restrict__ T_pred, int B,
                               float *__restrict__ rot_t, float *__restrict__ trans_t, float *__restrict__ tq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *a = res_r + b * 9, *r = R_pred + b * 9, *t = T_pred + b * 3;
  float m[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m[i * 3 + j] = a[i * 3 + 0] * r[0 * 3 + j] + a[i * 3 + 1] * r[1 * 3 + j] + a[i * 3 + 2] * r[2 * 3 + j];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float v = a[i * 3 + 0] * t[0] + a[i * 3 + 1] * t[1] + a[i * 3 + 2] * t[2] + res_t[b * 3 + i];
    trans_t[b * 3 + i] = v;
    if (tq) tq[b * 7 + i] = v;
  }
  const float eps = 1e-8f;
  const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
  const float trace = m00 + m11 + m22;
  float qx, qy, qz, qw;      // kornia order (x, y, z, w)
  if (trace > 0.f) {
    const float sq = sqrtf(trace + 1.f) * 2.f;
    qx = pose_safe_div(m21 - m12, sq); qy = pose_safe_div(m02 - m20, sq); qz = pose_safe_div(m10 - m01, sq); qw = 0.25f * sq;
  } else if (m00 > m11 && m00 > m22) {
    const float sq = sqrtf(1.f + m00 - m11 - m22 + eps) * 2.f;
    qx = 0.25f * sq; qy = pose_safe_div(m01 + m10, sq); qz = pose_safe_div(m02 + m20, sq); qw = pose_safe_div(m21 - m12, sq);
  } else if (m11 > m22) {
    const float sq = sqrtf(1.f + m11 - m00 - m22 + eps) * 2.f;
    qx = pose_safe_div(m01 + m10, sq); qy = 0.25f * sq; qz = pose_safe_div(m12 + m21, sq); qw = pose_safe_div(m02 - m20, sq);
  } else {
    const float sq = sqrtf(1.f + m22 - m00 - m11 + eps) * 2.f;
    qx = pose_safe_div(m02 + m20, sq); qy = pose_safe_div(m12 + m21, sq); qz = 0.25f * sq; qw = pose_safe_div(m10 - m01, sq);
  }
  // roll to (w, x, y, z) and multiply by sign(w) (torch.sign: 0 for w == 0)
  const float s = qw > 0.f ? 1.f : (qw < 0.f ? -1.f : 0.f);
  rot_t[b * 4 + 0] = qw * s;
  rot_t[b * 4 + 1] = qx * s;
  rot_t[b * 4 + 2] = qy * s;
  rot_t[b * 4 + 3] = qz * s;
  if (tq) {      // the [B,7] (t, q) rows the pyramid supervision reads (voxel_odom_net.py:747): no concatenation launch
    tq[b * 7 + 3] = qw * s;
    tq[b * 7 + 4] = qx * s;
    tq[b * 7 + 5] = qy * s;
    tq[b * 7 + 6] = qz * s;
  }
}

extern "C" int rslo_quat_to_rot(const float *q_wxyz, int B, float *R, void *stream) {
  RSLO_CHECK_ARG(q_wxyz && R && B >= 0, "rslo_quat_to_rot: bad arguments");
  if (B == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_quat_to_rot, dim3((unsigned)rslo_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, q_wxyz, B, R);
  RSLO_CHECK_LAUNCH("k_quat_to_rot");
  return RSLO_OK;
}

extern "C" int rslo_quat_to_rot_bwd(const float *q_wxyz, const float *gR, int B, float *gq, void *stream) {
  RSLO_CHECK_ARG(q_wxyz && gR && gq && B >= 0, "rslo_quat_to_rot_bwd: bad arguments");
  if (B == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_quat_to_rot_bwd, dim3((unsigned)rslo_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, q_wxyz, gR,
                     B, gq);
  RSLO_CHECK_LAUNCH("k_quat_to_rot_bwd");
  return RSLO_OK;
}

extern "C" int rslo_pose_targets_tq(const float *res_r, const float *res_t, const float *R_pred, const float *T_pred,
                                    int B, float *rot_targets_wxyz, float *trans_targets, float *tq /*[B,7] or NULL*/,
                                    void *stream) {
  RSLO_CHECK_ARG(res_r && res_t && R_pred && T_pred && rot_targets_wxyz && trans_targets && B >= 0,
                 "rslo_pose_targets: bad arguments");
  if (B == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_pose_targets, dim3((unsigned)rslo_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, res_r, res_t,
                     R_pred, T_pred, B, rot_targets_wxyz, trans_targets, tq);
  RSLO_CHECK_LAUNCH("k_pose_targets");
  return RSLO_OK;
}

extern "C" int rslo_pose_targets(const float *res_r, const float *res_t, const float *R_pred, const float *T_pred, int B,
                                 float *rot_targets_wxyz, float *trans_targets, void *stream) {
  return rslo_pose_targets_tq(res_r, res_t, R_pred, T_pred, B, rot_targets_wxyz, trans_targets, nullptr, stream);
}
