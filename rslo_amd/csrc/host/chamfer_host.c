/*
 * librslo_host.so -- the HOST entry points of thirdparty.chamfer_distance: cd.forward / cd.backward on CPU tensors.
 *
 * The reference module computes these two on the host (thirdparty/chamfer_distance/chamfer_distance.cpp:147-234: two
 * exhaustive nearest-neighbour scans, then the gradient scatter of both directions), selected at chamfer_distance.py:34,61
 * when the inputs are not CUDA tensors.  They are outside the training path (the loss uses the one-direction CUDA entry
 * point), but they are part of the module's surface, so a user of `ChamferDistance()` on CPU tensors finds them here.
 * Plain C, no HIP, no torch.  Device tensors never come here: they go to rslo_chamfer_nn / rslo_chamfer_grad.
 *
 * Arithmetic contract (what makes idx identical to the reference's host code and to the GPU kernel): the squared distance
 * is formed in fp32 as ((dx dx + dy dy) + dz dz) without contraction (build with -ffp-contract=off), compared with a strict
 * '<' scanning k upwards, so the LOWEST index wins ties.
 */
#include <stdint.h>
#include <stddef.h>

#define RSLO_HOST_API __attribute__((visibility("default")))

/* for every point of a [b, n, 3]: squared distance to and index of its nearest point of c [b, m, 3] */
static void nearest(int b, int n, int m, const float *a, const float *c, float *dist, int32_t *idx) {
  for (int i = 0; i < b; ++i) {
    const float *cb = c + (size_t)i * m * 3;
    for (int j = 0; j < n; ++j) {
      const float *p = a + ((size_t)i * n + j) * 3;
      float best = 0.f;
      int32_t at = 0;
      for (int k = 0; k < m; ++k) {
        const float dx = cb[3 * k] - p[0], dy = cb[3 * k + 1] - p[1], dz = cb[3 * k + 2] - p[2];
        const float d = dx * dx + dy * dy + dz * dz;
        if (k == 0 || d < best) {
          best = d;
          at = k;
        }
      }
      dist[(size_t)i * n + j] = best;
      idx[(size_t)i * n + j] = at;
    }
  }
}

/* cd.forward: both directions.  Returns 0, or -1 on bad arguments (m == 0 or n == 0 leaves nothing to search). */
RSLO_HOST_API int rslo_host_chamfer_forward(const float *xyz1 /*[b,n,3]*/, const float *xyz2 /*[b,m,3]*/, int b, int n, int m,
                                            float *dist1 /*[b,n]*/, float *dist2 /*[b,m]*/, int32_t *idx1 /*[b,n]*/,
                                            int32_t *idx2 /*[b,m]*/) {
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || b < 0 || n < 1 || m < 1) return -1;
  nearest(b, n, m, xyz1, xyz2, dist1, idx1);
  nearest(b, m, n, xyz2, xyz1, dist2, idx2);
  return 0;
}

/* one direction of the gradient: d(dist)/d(point) = 2 (p - q) on the point, -2 (p - q) on its partner */
static void scatter(int b, int n, int m, const float *a, const float *c, const float *gdist, const int32_t *idx, float *ga,
                    float *gc) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const size_t r = (size_t)i * n + j;
      const size_t q = (size_t)i * m + (size_t)idx[r];
      const float g = gdist[r] * 2;
      for (int t = 0; t < 3; ++t) {
        const float v = g * (a[3 * r + t] - c[3 * q + t]);
        ga[3 * r + t] += v;
        gc[3 * q + t] -= v;
      }
    }
}

/* cd.backward: gradxyz1 / gradxyz2 are overwritten (zeroed, then both directions added in the reference's order: per batch
 * element the xyz1 -> xyz2 pairs first, then the xyz2 -> xyz1 pairs) */
RSLO_HOST_API int rslo_host_chamfer_backward(const float *xyz1, const float *xyz2, int b, int n, int m, const float *graddist1,
                                             const float *graddist2, const int32_t *idx1, const int32_t *idx2,
                                             float *gradxyz1 /*[b,n,3]*/, float *gradxyz2 /*[b,m,3]*/) {
  if (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2 || b < 0 || n < 1 || m < 1)
    return -1;
  for (size_t e = 0; e < (size_t)b * n * 3; ++e) gradxyz1[e] = 0.f;
  for (size_t e = 0; e < (size_t)b * m * 3; ++e) gradxyz2[e] = 0.f;
  for (int i = 0; i < b; ++i) {
    const size_t o1 = (size_t)i * n, o2 = (size_t)i * m;
    scatter(1, n, m, xyz1 + 3 * o1, xyz2 + 3 * o2, graddist1 + o1, idx1 + o1, gradxyz1 + 3 * o1, gradxyz2 + 3 * o2);
    scatter(1, m, n, xyz2 + 3 * o2, xyz1 + 3 * o1, graddist2 + o2, idx2 + o2, gradxyz2 + 3 * o2, gradxyz1 + 3 * o1);
  }
  return 0;
}
