// One-direction chamfer / nearest-neighbour search for gfx950.
//
// Brute force, fp32-VALU bound (N*M distance evaluations on < 1 MB of data).  The target cloud
// is cut into S segments so that the launch has >> 256 workgroups even for one 31k-point pair;
// a workgroup stages a target tile in LDS as per-coordinate (even, odd) pairs and every lane scans it
// for QPT queries of its own with packed fp32 math (all lanes read the same LDS address: broadcast,
// conflict-free).  A second kernel merges the S partial results in segment order.
//
// Arithmetic is the CPU reference's (chamfer_distance.cpp:116-144): per-term fp32 products and
// sums, NO fma contraction, strict '<' (lowest index wins ties) -- dist and idx are bit-exact
// against that path.
#include "rslo_common.h"

#define CH_THREADS 256
#define CH_QPT 4
#define CH_TILE 2048      /* targets per LDS tile (24 KB as x/y/z pairs) */
#define CH_CHUNK 64       /* targets per argmin chunk */

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Distance of one query to two targets at once, in the reference's exact arithmetic: per-term fp32
// products and sums, no contraction.  The float2 form lets the compiler use v_pk_add/mul_f32 (the packed
// fp32 rate is what the 157 TF vector peak is quoted on).
__device__ __forceinline__ f32x2 dist2(f32x2 tx, f32x2 ty, f32x2 tz, float qx, float qy, float qz) {
#pragma clang fp contract(off)
  const f32x2 dx = tx - qx, dy = ty - qy, dz = tz - qz;
  return (dx * dx + dy * dy) + dz * dz;
}

// The hot loop only tracks the running MINIMUM per query (v_min3 over a pair of targets) and, once per
// 64-target chunk, which chunk improved it (strict '<': the earliest chunk wins).  The index is recovered
// afterwards by re-scanning that one chunk for the first target whose distance equals the minimum -- the
// same expression gives the same bits, so this is exactly "lowest index wins ties".
__global__ __launch_bounds__(CH_THREADS) void k_chamfer_part(const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, int N, int M,
                                                             int seg_len, int S, const int32_t *__restrict__ mcnt,
                                                             float *__restrict__ pdist,
                                                             int32_t *__restrict__ pidx) {
  __shared__ f32x2 tx[CH_TILE / 2], ty[CH_TILE / 2], tz[CH_TILE / 2];
  const int b = blockIdx.z, s = blockIdx.y;
  const int Mb = mcnt ? mcnt[b] : M;          // ragged batch: only the first Mb targets of pair b exist
  const int m0 = s * seg_len;
  const int m1 = (m0 + seg_len < Mb) ? m0 + seg_len : Mb;
  const float *q = xyz1 + (int64_t)b * N * 3;
  const float *t = xyz2 + (int64_t)b * M * 3;
  const int q0 = blockIdx.x * (CH_THREADS * CH_QPT) + threadIdx.x;
  const float INF = __builtin_inff();

  float qx[CH_QPT], qy[CH_QPT], qz[CH_QPT], best[CH_QPT];
  int bchunk[CH_QPT];
#pragma unroll
  for (int j = 0; j < CH_QPT; ++j) {
    const int qi = q0 + j * CH_THREADS;
    const bool ok = qi < N;
    qx[j] = ok ? q[qi * 3 + 0] : 0.f;
    qy[j] = ok ? q[qi * 3 + 1] : 0.f;
    qz[j] = ok ? q[qi * 3 + 2] : 0.f;
    best[j] = INF;
    bchunk[j] = m0;
  }

  for (int k0 = m0; k0 < m1; k0 += CH_TILE) {
    const int cnt = (m1 - k0 < CH_TILE) ? (m1 - k0) : CH_TILE;
    __syncthreads();
    // stage the tile as (even, odd) target pairs per coordinate; pad the tail with +inf coordinates
    for (int e = threadIdx.x; e < CH_TILE / 2; e += CH_THREADS) {
      const int a = 2 * e, c = 2 * e + 1;
      const float *pa = t + (int64_t)(k0 + a) * 3, *pc = t + (int64_t)(k0 + c) * 3;
      const bool va = a < cnt, vc = c < cnt;
      tx[e] = (f32x2){va ? pa[0] : INF, vc ? pc[0] : INF};
      ty[e] = (f32x2){va ? pa[1] : INF, vc ? pc[1] : INF};
      tz[e] = (f32x2){va ? pa[2] : INF, vc ? pc[2] : INF};
    }
    __syncthreads();
    const int nchunk = (cnt + CH_CHUNK - 1) / CH_CHUNK;
    for (int c = 0; c < nchunk; ++c) {
      float cmin[CH_QPT];
#pragma unroll
      for (int j = 0; j < CH_QPT; ++j) cmin[j] = INF;
#pragma unroll 8
      for (int e = c * (CH_CHUNK / 2); e < (c + 1) * (CH_CHUNK / 2); ++e) {
        const f32x2 x = tx[e], y = ty[e], z = tz[e];
#pragma unroll
        for (int j = 0; j < CH_QPT; ++j) {
          const f32x2 d = dist2(x, y, z, qx[j], qy[j], qz[j]);
          cmin[j] = fminf(cmin[j], fminf(d.x, d.y));
        }
      }
#pragma unroll
      for (int j = 0; j < CH_QPT; ++j)
        if (cmin[j] < best[j]) {
          best[j] = cmin[j];
          bchunk[j] = k0 + c * CH_CHUNK;
        }
    }
  }

  // recover the index: first target of the winning chunk with d == best (straight from global memory)
#pragma unroll
  for (int j = 0; j < CH_QPT; ++j) {
    const int qi = q0 + j * CH_THREADS;
    if (qi >= N) continue;
    int bi = m0;
    if (best[j] < INF) {
      const int c0 = bchunk[j];
      const int c1 = (c0 + CH_CHUNK < m1) ? c0 + CH_CHUNK : m1;
      bi = c0;
      for (int k = c1 - 1; k >= c0; --k) {
#pragma clang fp contract(off)
        const float dx = t[(int64_t)k * 3 + 0] - qx[j], dy = t[(int64_t)k * 3 + 1] - qy[j],
                    dz = t[(int64_t)k * 3 + 2] - qz[j];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d == best[j]) bi = k;
      }
    }
    pdist[((int64_t)b * S + s) * N + qi] = best[j];
    pidx[((int64_t)b * S + s) * N + qi] = bi;
  }
}

__global__ void k_chamfer_merge(const float *__restrict__ pdist, const int32_t *__restrict__ pidx, int N, int S,
                                const int32_t *__restrict__ ncnt, float *__restrict__ dist,
                                int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (ncnt && i >= ncnt[b]) {   // padding row of a ragged batch: never inside any ROI
    dist[(int64_t)b * N + i] = __builtin_inff();
    idx[(int64_t)b * N + i] = 0;
    return;
  }
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
  int S = (int)rslo_cdiv(1024, qblocks);  // aim at ~1k workgroups (4 per CU)
  const int maxS = (int)rslo_cdiv(M > 0 ? M : 1, 256);
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  return S;
}

extern "C" size_t rslo_chamfer_grid_ws_bytes(int B, int N, int M);
extern "C" int rslo_chamfer_brute_nn(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *ncnt,
                                     const int32_t *mcnt, float *dist, int32_t *idx, void *ws, size_t ws_bytes,
                                     void *stream);
extern "C" int rslo_chamfer_grid_nn(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *ncnt,
                                    const int32_t *mcnt, float *dist, int32_t *idx, void *ws, size_t ws_bytes,
                                    void *stream);

// which search runs behind rslo_chamfer_nn*: the pruned one once the clouds are large enough to amortise its
// five small launches (same results either way); the `chamfer` switch forces one (1 exhaustive, 2 pruned)
static bool chamfer_use_grid(int N, int M) {
  const int force = rslo_tune(RSLO_TUNE_CHAMFER);
  if (force == 1) return false;
  if (force == 2) return true;
  return N >= 1024 && M >= 2048;
}

static size_t chamfer_brute_ws_bytes(int B, int N, int M) {
  return (size_t)chamfer_segments(B, N, M) * (size_t)(B > 0 ? B : 1) * (size_t)(N > 0 ? N : 1) * 8;
}

extern "C" size_t rslo_chamfer_ws_bytes(int B, int N, int M) {
  const size_t a = chamfer_brute_ws_bytes(B, N, M), g = rslo_chamfer_grid_ws_bytes(B, N, M);
  return a > g ? a : g;
}

extern "C" int rslo_chamfer_nn_ragged(const float *xyz1, const float *xyz2, int B, int N, int M,
                                      const int32_t *ncnt, const int32_t *mcnt, float *dist, int32_t *idx,
                                      void *ws, size_t ws_bytes, void *stream);

extern "C" int rslo_chamfer_nn(const float *xyz1, const float *xyz2, int B, int N, int M, float *dist,
                               int32_t *idx, void *ws, size_t ws_bytes, void *stream) {
  return rslo_chamfer_nn_ragged(xyz1, xyz2, B, N, M, nullptr, nullptr, dist, idx, ws, ws_bytes, stream);
}

extern "C" int rslo_chamfer_nn_ragged(const float *xyz1, const float *xyz2, int B, int N, int M,
                                      const int32_t *ncnt, const int32_t *mcnt, float *dist, int32_t *idx,
                                      void *ws, size_t ws_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(B >= 0 && N >= 0 && M >= 1, "chamfer_nn: need M >= 1");
  if (B == 0 || N == 0) return RSLO_OK;
  if (chamfer_use_grid(N, M)) return rslo_chamfer_grid_nn(xyz1, xyz2, B, N, M, ncnt, mcnt, dist, idx, ws, ws_bytes, stream);
  return rslo_chamfer_brute_nn(xyz1, xyz2, B, N, M, ncnt, mcnt, dist, idx, ws, ws_bytes, stream);
}

extern "C" int rslo_chamfer_brute_nn(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *ncnt,
                                     const int32_t *mcnt, float *dist, int32_t *idx, void *ws, size_t ws_bytes,
                                     void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(B >= 0 && N >= 0 && M >= 1, "chamfer_nn: need M >= 1");
  if (B == 0 || N == 0) return RSLO_OK;
  const int S = chamfer_segments(B, N, M);
  if (ws_bytes < chamfer_brute_ws_bytes(B, N, M)) {
    rslo_set_error("chamfer_nn: workspace too small");
    return RSLO_EWS;
  }
  const int seg_len = (int)rslo_cdiv(M, S);
  float *pdist = (float *)ws;
  int32_t *pidx = (int32_t *)(pdist + (int64_t)S * B * N);
  dim3 grid((unsigned)rslo_cdiv(N, CH_THREADS * CH_QPT), (unsigned)S, (unsigned)B);
  hipLaunchKernelGGL(k_chamfer_part, grid, dim3(CH_THREADS), 0, st, xyz1, xyz2, N, M, seg_len, S, mcnt, pdist,
                     pidx);
  hipLaunchKernelGGL(k_chamfer_merge, dim3((unsigned)rslo_cdiv(N, 256), (unsigned)B), dim3(256), 0, st, pdist,
                     pidx, N, S, ncnt, dist, idx);
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
