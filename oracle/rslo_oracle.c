/*
 * rslo_oracle.c -- CPU restatement of the RSLO two-frame odometry hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity checker for the HIP
 * path and the "port" CPU baseline of bench.py.  Nothing in the product
 * package (rslo_amd/) may import, link or call it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - orc_chamfer_nn       : PINNED. Restates thirdparty/chamfer_distance/
 *                            chamfer_distance.cpp:116-144 (nnsearch) and is
 *                            checked bit-for-bit against oracle/_ref (the
 *                            reference .cpp compiled from where it lies) and
 *                            against tests/golden/chamfer_*.npz.
 *   - orc_voxelize, orc_rulebook_*, orc_spconv_*, orc_dense
 *                          : PARITY UNPINNED by any reference source: the
 *                            reference only *calls* spconv (DecaYale/spconv_plus,
 *                            un-pinned HEAD, Dockerfile:55-61; call sites
 *                            rslo/models/middle.py:80-97,119-213,224,240 and
 *                            rslo/builder/voxel_builder.py:36-54,83-94).  These
 *                            functions restate the published spconv-1.x /
 *                            SECOND contract and are pinned instead by the
 *                            dense F.conv3d / conv_transpose3d equivalence
 *                            (tests/test_oracle.py) and hand-made KATs.
 *
 * Conventions: coordinates are (b, z, y, x) int32; weights are
 * [K = kz*ky*kx, Cin, Cout] row-major with k = (kz*KY + ky)*KX + kx
 * (view of spconv's [kz,ky,kx,Cin,Cout]); a "neighbour table" nbr[M*K]
 * holds for output row o and kernel offset k the contributing input row
 * or -1.  Output rows of a strided conv are numbered by ascending linear
 * index ((b*D+z)*H+y)*W+x (what spconv-1.x's CUDA path produces through
 * sort+unique); SubM and inverse conv keep the given input order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* a1: voxelization.  Classic SECOND/spconv-1.0 points_to_voxel loop:   */
/* first-come voxel numbering, <=T points per voxel, stop at the first  */
/* point that would open voxel number max_voxels (`break`), coords zyx. */
/* float32 arithmetic for floor((p - lo) / vs), as the templated C++    */
/* (DType=float) and the numba kernel both do.                          */
/* Call site: rslo/data/preprocess.py:493 via voxel_builder.py:48-54.   */
/* ------------------------------------------------------------------ */
ORC_API int64_t orc_voxelize(const float *pts, int64_t P, int F,
                             const float *range6, const float *vsize3,
                             const int32_t *grid_xyz, int T, int max_voxels,
                             float *voxels, int32_t *coords,
                             int32_t *num_pts) {
  const int64_t gx = grid_xyz[0], gy = grid_xyz[1], gz = grid_xyz[2];
  int32_t *lut = (int32_t *)malloc(sizeof(int32_t) * gx * gy * gz);
  if (!lut) return -1;
  memset(lut, 0xFF, sizeof(int32_t) * gx * gy * gz);
  memset(voxels, 0, sizeof(float) * (size_t)max_voxels * T * F);
  memset(num_pts, 0, sizeof(int32_t) * (size_t)max_voxels);
  int64_t nvox = 0;
  for (int64_t i = 0; i < P; ++i) {
    int32_t c[3];
    int ok = 1;
    for (int j = 0; j < 3; ++j) {
      volatile float d = pts[i * F + j] - range6[j];
      volatile float q = d / vsize3[j];
      float fl = floorf(q);
      if (!(fl >= 0.0f) || !(fl < (float)grid_xyz[j])) { ok = 0; break; }
      c[j] = (int32_t)fl;
    }
    if (!ok) continue;
    int64_t lin = ((int64_t)c[2] * gy + c[1]) * gx + c[0];
    int32_t v = lut[lin];
    if (v < 0) {
      if (nvox >= max_voxels) break;
      v = (int32_t)nvox++;
      lut[lin] = v;
      coords[v * 3 + 0] = c[2];
      coords[v * 3 + 1] = c[1];
      coords[v * 3 + 2] = c[0];
    }
    int32_t n = num_pts[v];
    if (n < T) {
      memcpy(voxels + ((size_t)v * T + n) * F, pts + i * F, sizeof(float) * F);
      num_pts[v] = n + 1;
    }
  }
  free(lut);
  return nvox;
}

/* ------------------------------------------------------------------ */
/* a5: rulebooks.                                                       */
/* ------------------------------------------------------------------ */
static inline int64_t lin_idx(int b, int z, int y, int x, const int32_t *d) {
  return (((int64_t)b * d[0] + z) * d[1] + y) * d[2] + x;
}

static int32_t *build_lut(const int32_t *coords, int64_t N, int B,
                          const int32_t *dims) {
  int64_t vol = (int64_t)B * dims[0] * dims[1] * dims[2];
  int32_t *lut = (int32_t *)malloc(sizeof(int32_t) * vol);
  if (!lut) return NULL;
  memset(lut, 0xFF, sizeof(int32_t) * vol);
  for (int64_t i = 0; i < N; ++i) {
    const int32_t *c = coords + i * 4;
    lut[lin_idx(c[0], c[1], c[2], c[3], dims)] = (int32_t)i;
  }
  return lut;
}

/* SubM conv (spconv.SubMConv3d, middle.py:121-210): output set == input set,
 * same order; nbr[o][k] = row at coords[o] + (k - ks/2). */
ORC_API int orc_rulebook_subm(const int32_t *coords, int64_t N, int B,
                              const int32_t *dims, const int32_t *ks,
                              int32_t *nbr) {
  int32_t *lut = build_lut(coords, N, B, dims);
  if (!lut) return -1;
  const int K = ks[0] * ks[1] * ks[2];
  for (int64_t o = 0; o < N; ++o) {
    const int32_t *c = coords + o * 4;
    int k = 0;
    for (int kz = 0; kz < ks[0]; ++kz)
      for (int ky = 0; ky < ks[1]; ++ky)
        for (int kx = 0; kx < ks[2]; ++kx, ++k) {
          int z = c[1] + kz - ks[0] / 2, y = c[2] + ky - ks[1] / 2,
              x = c[3] + kx - ks[2] / 2;
          int32_t r = -1;
          if (z >= 0 && z < dims[0] && y >= 0 && y < dims[1] && x >= 0 &&
              x < dims[2])
            r = lut[lin_idx(c[0], z, y, x, dims)];
          nbr[o * K + k] = r;
        }
  }
  free(lut);
  return 0;
}

/* Strided SparseConv3d (middle.py:127-169): an output site exists iff some
 * active input lies in its receptive field: in = out*stride - pad + k.
 * Writes the output coordinates in ascending linear order; returns M. */
ORC_API int64_t orc_conv_out_coords(const int32_t *coords, int64_t N, int B,
                                    const int32_t *in_dims, const int32_t *ks,
                                    const int32_t *stride, const int32_t *pad,
                                    const int32_t *out_dims,
                                    int32_t *out_coords, int64_t cap) {
  int64_t vol = (int64_t)B * out_dims[0] * out_dims[1] * out_dims[2];
  uint8_t *mark = (uint8_t *)calloc((size_t)vol, 1);
  if (!mark) return -1;
  (void)in_dims;
  for (int64_t i = 0; i < N; ++i) {
    const int32_t *c = coords + i * 4;
    for (int kz = 0; kz < ks[0]; ++kz) {
      int tz = c[1] + pad[0] - kz;
      if (tz < 0 || tz % stride[0]) continue;
      int z = tz / stride[0];
      if (z >= out_dims[0]) continue;
      for (int ky = 0; ky < ks[1]; ++ky) {
        int ty = c[2] + pad[1] - ky;
        if (ty < 0 || ty % stride[1]) continue;
        int y = ty / stride[1];
        if (y >= out_dims[1]) continue;
        for (int kx = 0; kx < ks[2]; ++kx) {
          int tx = c[3] + pad[2] - kx;
          if (tx < 0 || tx % stride[2]) continue;
          int x = tx / stride[2];
          if (x >= out_dims[2]) continue;
          mark[lin_idx(c[0], z, y, x, out_dims)] = 1;
        }
      }
    }
  }
  int64_t M = 0;
  const int64_t hw = (int64_t)out_dims[1] * out_dims[2];
  for (int64_t l = 0; l < vol; ++l)
    if (mark[l]) {
      if (M < cap) {
        int64_t r = l;
        int x = (int)(r % out_dims[2]); r /= out_dims[2];
        int y = (int)(r % out_dims[1]); r /= out_dims[1];
        int z = (int)(r % out_dims[0]); r /= out_dims[0];
        out_coords[M * 4 + 0] = (int32_t)r;
        out_coords[M * 4 + 1] = z;
        out_coords[M * 4 + 2] = y;
        out_coords[M * 4 + 3] = x;
      }
      ++M;
    }
  (void)hw;
  free(mark);
  return M;
}

/* nbr[o][k]  = input row at out*stride - pad + k           (forward gather)
 * nbrT[i][k] = output row o with o*stride - pad + k == in  (dgrad / inverse) */
ORC_API int orc_conv_tables(const int32_t *in_coords, int64_t N,
                            const int32_t *out_coords, int64_t M, int B,
                            const int32_t *in_dims, const int32_t *ks,
                            const int32_t *stride, const int32_t *pad,
                            const int32_t *out_dims, int32_t *nbr,
                            int32_t *nbrT) {
  int32_t *lut = build_lut(in_coords, N, B, in_dims);
  if (!lut) return -1;
  const int K = ks[0] * ks[1] * ks[2];
  if (nbrT)
    for (int64_t i = 0; i < N * K; ++i) nbrT[i] = -1;
  for (int64_t o = 0; o < M; ++o) {
    const int32_t *c = out_coords + o * 4;
    int k = 0;
    for (int kz = 0; kz < ks[0]; ++kz)
      for (int ky = 0; ky < ks[1]; ++ky)
        for (int kx = 0; kx < ks[2]; ++kx, ++k) {
          int z = c[1] * stride[0] - pad[0] + kz;
          int y = c[2] * stride[1] - pad[1] + ky;
          int x = c[3] * stride[2] - pad[2] + kx;
          int32_t r = -1;
          if (z >= 0 && z < in_dims[0] && y >= 0 && y < in_dims[1] && x >= 0 &&
              x < in_dims[2])
            r = lut[lin_idx(c[0], z, y, x, in_dims)];
          nbr[o * K + k] = r;
          if (r >= 0 && nbrT) nbrT[(int64_t)r * K + k] = (int32_t)o;
        }
  }
  (void)out_dims;
  free(lut);
  return 0;
}

/* spconv-1.x style export: indice_pairs[K][2][P_k] laid out with row stride
 * `cap`, indice_num[K]; pairs of offset k listed in ascending INPUT row
 * (the order of spconv's CPU getIndicePairsConv loop over inputs). */
ORC_API int orc_pairs_from_nbrT(const int32_t *nbrT, int64_t N, int K,
                                int64_t cap, int32_t *pairs,
                                int32_t *indice_num) {
  for (int k = 0; k < K; ++k) indice_num[k] = 0;
  for (int64_t i = 0; i < N; ++i)
    for (int k = 0; k < K; ++k) {
      int32_t o = nbrT[i * K + k];
      if (o < 0) continue;
      int32_t n = indice_num[k]++;
      if (n < cap) {
        pairs[((int64_t)k * 2 + 0) * cap + n] = (int32_t)i;
        pairs[((int64_t)k * 2 + 1) * cap + n] = o;
      }
    }
  return 0;
}

/* ------------------------------------------------------------------ */
/* a6: sparse conv arithmetic  Y[o] = bias + sum_k X[nbr[o][k]] W_k.    */
/* Double accumulation -> rounded once to fp32: the checker is more     */
/* accurate than either the reference's cuBLAS fp32 or our MFMA fp32.   */
/* ------------------------------------------------------------------ */
ORC_API int orc_spconv_fwd(const float *in, int Cin, const float *W,
                           const float *bias, const int32_t *nbr, int64_t M,
                           int K, int Cout, float *out) {
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < M; ++o) {
    double acc[256];
    for (int c = 0; c < Cout; ++c) acc[c] = bias ? (double)bias[c] : 0.0;
    for (int k = 0; k < K; ++k) {
      int32_t i = nbr[o * K + k];
      if (i < 0) continue;
      const float *x = in + (int64_t)i * Cin;
      const float *w = W + (int64_t)k * Cin * Cout;
      for (int ci = 0; ci < Cin; ++ci) {
        double xv = x[ci];
        const float *wr = w + ci * Cout;
        for (int c = 0; c < Cout; ++c) acc[c] += xv * (double)wr[c];
      }
    }
    for (int c = 0; c < Cout; ++c) out[o * Cout + c] = (float)acc[c];
  }
  return 0;
}

/* dX[i] = sum_k dY[nbrT[i][k]] W_k^T   (gather form over the transposed table) */
ORC_API int orc_spconv_dgrad(const float *dout, int Cout, const float *W,
                             const int32_t *nbrT, int64_t N, int K, int Cin,
                             float *din) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    double acc[256];
    for (int c = 0; c < Cin; ++c) acc[c] = 0.0;
    for (int k = 0; k < K; ++k) {
      int32_t o = nbrT[i * K + k];
      if (o < 0) continue;
      const float *g = dout + (int64_t)o * Cout;
      const float *w = W + (int64_t)k * Cin * Cout;
      for (int ci = 0; ci < Cin; ++ci) {
        const float *wr = w + ci * Cout;
        double s = 0.0;
        for (int c = 0; c < Cout; ++c) s += (double)g[c] * (double)wr[c];
        acc[ci] += s;
      }
    }
    for (int c = 0; c < Cin; ++c) din[i * Cin + c] = (float)acc[c];
  }
  return 0;
}

/* dW_k = sum_o X[nbr[o][k]]^T dY[o] ;  db = sum_o dY[o] */
ORC_API int orc_spconv_wgrad(const float *in, int Cin, const float *dout,
                             int Cout, const int32_t *nbr, int64_t M, int K,
                             float *dW, float *dbias) {
#pragma omp parallel for schedule(dynamic)
  for (int k = 0; k < K; ++k) {
    double *acc = (double *)calloc((size_t)Cin * Cout, sizeof(double));
    for (int64_t o = 0; o < M; ++o) {
      int32_t i = nbr[o * K + k];
      if (i < 0) continue;
      const float *x = in + (int64_t)i * Cin;
      const float *g = dout + o * Cout;
      for (int ci = 0; ci < Cin; ++ci) {
        double xv = x[ci];
        double *a = acc + ci * Cout;
        for (int c = 0; c < Cout; ++c) a[c] += xv * (double)g[c];
      }
    }
    float *d = dW + (int64_t)k * Cin * Cout;
    for (int j = 0; j < Cin * Cout; ++j) d[j] = (float)acc[j];
    free(acc);
  }
  if (dbias) {
    for (int c = 0; c < Cout; ++c) {
      double s = 0.0;
      for (int64_t o = 0; o < M; ++o) s += dout[o * Cout + c];
      dbias[c] = (float)s;
    }
  }
  return 0;
}

/* a8: SparseConvTensor.dense() -> [B, C, D, H, W] (middle.py:240). */
ORC_API int orc_dense(const float *feat, const int32_t *coords, int64_t M,
                      int C, int B, const int32_t *dims, float *out) {
  int64_t vol = (int64_t)dims[0] * dims[1] * dims[2];
  memset(out, 0, sizeof(float) * (size_t)B * C * vol);
  for (int64_t r = 0; r < M; ++r) {
    const int32_t *c = coords + r * 4;
    int64_t sp = ((int64_t)c[1] * dims[1] + c[2]) * dims[2] + c[3];
    for (int ch = 0; ch < C; ++ch)
      out[((int64_t)c[0] * C + ch) * vol + sp] = feat[r * C + ch];
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* a17: one-direction chamfer / nearest neighbour.                      */
/* Restates chamfer_distance.cpp:116-144 (nnsearch): per-term fp32      */
/* products and sums (no contraction), comparison on the widened value, */
/* strict `<` => lowest index wins ties; k==0 seeds the search.         */
/* Compile this file with -ffp-contract=off (the Makefile does).        */
/* ------------------------------------------------------------------ */
ORC_API int orc_chamfer_nn(int b, int n, int m, const float *xyz1,
                           const float *xyz2, float *dist, int32_t *idx) {
  for (int i = 0; i < b; ++i) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) {
      const float x1 = xyz1[((int64_t)i * n + j) * 3 + 0];
      const float y1 = xyz1[((int64_t)i * n + j) * 3 + 1];
      const float z1 = xyz1[((int64_t)i * n + j) * 3 + 2];
      double best = 0;
      int besti = 0;
      for (int k = 0; k < m; ++k) {
        const float x2 = xyz2[((int64_t)i * m + k) * 3 + 0] - x1;
        const float y2 = xyz2[((int64_t)i * m + k) * 3 + 1] - y1;
        const float z2 = xyz2[((int64_t)i * m + k) * 3 + 2] - z1;
        const float df = x2 * x2 + y2 * y2 + z2 * z2;
        const double d = df;
        if (k == 0 || d < best) { best = d; besti = k; }
      }
      dist[(int64_t)i * n + j] = (float)best;
      idx[(int64_t)i * n + j] = besti;
    }
  }
  return 0;
}

/* chamfer_distance.cpp:170-234 restricted to direction 1 (the one-direction
 * variant the configured loss uses, chamfer_distance.cu:177-206): g = 2*gd. */
ORC_API int orc_chamfer_grad(int b, int n, int m, const float *xyz1,
                             const float *xyz2, const float *graddist1,
                             const int32_t *idx1, float *g1, float *g2) {
  memset(g1, 0, sizeof(float) * (size_t)b * n * 3);
  memset(g2, 0, sizeof(float) * (size_t)b * m * 3);
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const float *p = xyz1 + ((int64_t)i * n + j) * 3;
      int j2 = idx1[(int64_t)i * n + j];
      const float *q = xyz2 + ((int64_t)i * m + j2) * 3;
      float g = graddist1[(int64_t)i * n + j] * 2;
      for (int a = 0; a < 3; ++a) {
        g1[((int64_t)i * n + j) * 3 + a] += g * (p[a] - q[a]);
        g2[((int64_t)i * m + j2) * 3 + a] -= g * (p[a] - q[a]);
      }
    }
  return 0;
}

ORC_API int orc_abi_version(void) { return 1; }
