// One-direction chamfer / nearest-neighbour search for gfx950.
//
// Brute force, fp32-VALU bound (N*M distance evaluations on < 1 MB of data).  The target cloud
// is cut into S segments so that the launch has >> 256 workgroups even for one 31k-point pair;
// a workgroup stages a target tile in LDS as float4 and every lane scans it for QPT queries of
// its own (all lanes read the same LDS address: broadcast, conflict-free).  A second kernel
// merges the S partial results in segment order.
//
// Arithmetic is the CPU reference's (chamfer_distance.cpp:116-144): per-term fp32 products and
// sums, NO fma contraction, strict '<' (lowest index wins ties) -- dist and idx are bit-exact
// against that path.
#include "rslo_common.h"

#define CH_THREADS 256
#define CH_QPT 4
#define CH_TILE 1024

__global__ __launch_bounds__(CH_THREADS) void k_chamfer_part(const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, int N, int M,
                                                             int seg_len, int S, float *__restrict__ pdist,
                                                             int32_t *__restrict__ pidx) {
#pragma clang fp contract(off)
  __shared__ float4 tile[CH_TILE];
  const int b = blockIdx.z, s = blockIdx.y;
  const int m0 = s * seg_len;
  const int m1 = (m0 + seg_len < M) ? m0 + seg_len : M;
  const float *q = xyz1 + (int64_t)b * N * 3;
  const float *t = xyz2 + (int64_t)b * M * 3;
  const int q0 = blockIdx.x * (CH_THREADS * CH_QPT) + threadIdx.x;

  float qx[CH_QPT], qy[CH_QPT], qz[CH_QPT], best[CH_QPT];
  int besti[CH_QPT];
#pragma unroll
  for (int j = 0; j < CH_QPT; ++j) {
    const int qi = q0 + j * CH_THREADS;
    const bool ok = qi < N;
    qx[j] = ok ? q[qi * 3 + 0] : 0.f;
    qy[j] = ok ? q[qi * 3 + 1] : 0.f;
    qz[j] = ok ? q[qi * 3 + 2] : 0.f;
    best[j] = __builtin_inff();
    besti[j] = m0;
  }

  for (int k0 = m0; k0 < m1; k0 += CH_TILE) {
    const int cnt = (m1 - k0 < CH_TILE) ? (m1 - k0) : CH_TILE;
    __syncthreads();
    for (int e = threadIdx.x; e < cnt; e += CH_THREADS) {
      const float *p = t + (int64_t)(k0 + e) * 3;
      tile[e] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < cnt; ++k) {
      const float4 p = tile[k];
#pragma unroll
      for (int j = 0; j < CH_QPT; ++j) {
        const float dx = p.x - qx[j], dy = p.y - qy[j], dz = p.z - qz[j];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < best[j]) {
          best[j] = d;
          besti[j] = k0 + k;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CH_QPT; ++j) {
    const int qi = q0 + j * CH_THREADS;
    if (qi < N) {
      pdist[((int64_t)b * S + s) * N + qi] = best[j];
      pidx[((int64_t)b * S + s) * N + qi] = besti[j];
    }
  }
}

__global__ void k_chamfer_merge(const float *__restrict__ pdist, const int32_t *__restrict__ pidx, int N, int S,
                                float *__restrict__ dist, int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float best = pdist[((int64_t)b * S) * N + i];
  int besti = pidx[((int64_t)b * S) * N + i];
  for (int s = 1; s < S; ++s) {
    const float d = pdist[((int64_t)b * S + s) * N + i];
    if (d < best) {
      best = d;
      besti = pidx[((int64_t)b * S + s) * N + i];
    }
  }
  // the reference seeds the search with candidate 0 (k==0 branch): with no finite distance
  // (NaN input) it answers index 0 / its distance; inf stays inf at index 0.
  dist[(int64_t)b * N + i] = best;
  idx[(int64_t)b * N + i] = besti;
}

static int chamfer_segments(int B, int N, int M) {
  const int64_t qblocks = rslo_cdiv(N > 0 ? N : 1, CH_THREADS * CH_QPT) * (B > 0 ? B : 1);
  int S = (int)rslo_cdiv(2048, qblocks);  // aim at ~2k workgroups (8 per CU)
  const int maxS = (int)rslo_cdiv(M > 0 ? M : 1, 256);
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  return S;
}

extern "C" size_t rslo_chamfer_ws_bytes(int B, int N, int M) {
  return (size_t)chamfer_segments(B, N, M) * (size_t)(B > 0 ? B : 1) * (size_t)(N > 0 ? N : 1) * 8;
}

extern "C" int rslo_chamfer_nn(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist,
                               int32_t *idx, void *ws, size_t ws_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(B >= 0 && N >= 0 && M >= 1, "chamfer_nn: need M >= 1");
  if (B == 0 || N == 0) return RSLO_OK;
  const int S = chamfer_segments(B, N, M);
  if (ws_bytes < rslo_chamfer_ws_bytes(B, N, M)) {
    rslo_set_error("chamfer_nn: workspace too small");
    return RSLO_EWS;
  }
  const int seg_len = (int)rslo_cdiv(M, S);
  float *pdist = (float *)ws;
  int32_t *pidx = (int32_t *)(pdist + (int64_t)S * B * N);
  dim3 grid((unsigned)rslo_cdiv(N, CH_THREADS * CH_QPT), (unsigned)S, (unsigned)B);
  hipLaunchKernelGGL(k_chamfer_part, grid, dim3(CH_THREADS), 0, st, xyz1, xyz2, N, M, seg_len, S, pdist,
                     pidx);
  hipLaunchKernelGGL(k_chamfer_merge, dim3((unsigned)rslo_cdiv(N, 256), (unsigned)B), dim3(256), 0, st, pdist,
                     pidx, N, S, dist, idx);
  RSLO_CHECK_LAUNCH("chamfer_nn");
  return RSLO_OK;
}

// chamfer_distance.cu:177-206 (one-direction): g = 2*graddist1; g1[i] += g (p - q); g2[idx] -= g (p - q)
__global__ void k_chamfer_grad(const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int M,
                               const float *__restrict__ gd, const int32_t *__restrict__ idx,
                               float *__restrict__ g1, float *__restrict__ g2) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float *p = xyz1 + ((int64_t)b * N + i) * 3;
  const int j2 = idx[(int64_t)b * N + i];
  const float *q = xyz2 + ((int64_t)b * M + j2) * 3;
  const float g = gd[(int64_t)b * N + i] * 2.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = g * (p[a] - q[a]);
    g1[((int64_t)b * N + i) * 3 + a] = v;
    atomicAdd(&g2[((int64_t)b * M + j2) * 3 + a], -v);
  }
}

extern "C" int rslo_chamfer_grad(const float *xyz1, const float *xyz2, int B, int N, int M,
                                 const float *graddist1, const int32_t *idx1, float *gradxyz1,
                                 float *gradxyz2, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_HIP(hipMemsetAsync(gradxyz2, 0, (size_t)B * M * 3 * sizeof(float), st));
  if (B == 0 || N == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_chamfer_grad, dim3((unsigned)rslo_cdiv(N, 256), (unsigned)B), dim3(256), 0, st, xyz1,
                     xyz2, N, M, graddist1, idx1, gradxyz1, gradxyz2);
  RSLO_CHECK_LAUNCH("chamfer_grad");
  return RSLO_OK;
}
