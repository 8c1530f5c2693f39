/*
 * librslo_host.so -- the HOST half of the operator boundary: spconv.utils.VoxelGenerator.generate on numpy arrays.
 *
 * The reference voxelizes on the CPU inside forked DataLoader workers (rslo/data/preprocess.py:493 through
 * rslo/builder/voxel_builder.py:36-54,83-94; train_hdf5.py:549-553 starts the workers with the default "fork").  A
 * forked child of a process that has initialised HIP cannot touch the device, so the numpy face of VoxelGenerator needs
 * an implementation that lives entirely in host memory.  This is that implementation: plain C, no HIP, no torch,
 * re-entrant (every call owns its scratch), so any number of workers can run it side by side.  CUDA tensors never come
 * here -- they are voxelized by rslo_voxelize / rslo_plan_encoder on the GPU.
 *
 * Semantics = rslo_voxelize (include/rslo_hip.h), bit for bit (tests/test_gpu_kernels.py compares the two on full
 * scans): cell = floor((p - lo) / vs) per axis in fp32 (no contraction: build with -ffp-contract=off), points outside
 * the grid skipped, voxels numbered in order of first appearance, the first T points of a voxel kept in point order,
 * processing stops at the first point that would open voxel number max_voxels, coordinates stored (z, y, x).
 *
 * Layout: instead of the dense z*y*x lookup table of the classic SECOND loop (177 MB of int32 for the shipped grid,
 * cleared on every call in every worker) the voxel ids live in an open-addressing table of 2 * max_voxels slots
 * (640 KB for 40000 voxels: L2-resident on the host), keyed by the linear cell index.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RSLO_HOST_API __attribute__((visibility("default")))

RSLO_HOST_API int rslo_host_abi_version(void) { return 1; }

/* returns the number of voxels (>= 0), -1 on bad arguments, -2 when scratch memory cannot be had.
 * voxels [max_voxels,T,F], coords [max_voxels,3], num_points [max_voxels]: caller-owned, need NOT be initialised --
 * rows [0, return value) are written completely (unused point slots zero), rows past it are left untouched. */
RSLO_HOST_API int64_t rslo_host_voxelize(const float *points, int64_t P, int F, const float *range6,
                                         const float *vsize3, const int32_t *grid_xyz, int T, int max_voxels,
                                         float *voxels, int32_t *coords, int32_t *num_points) {
  if (!points || !range6 || !vsize3 || !grid_xyz || !voxels || !coords || !num_points) return -1;
  if (P < 0 || F < 3 || T < 1 || max_voxels < 1) return -1;
  const uint64_t gx = (uint64_t)grid_xyz[0], gy = (uint64_t)grid_xyz[1], gz = (uint64_t)grid_xyz[2];
  if (grid_xyz[0] < 1 || grid_xyz[1] < 1 || grid_xyz[2] < 1 || gx * gy * gz >= 0xFFFFFFFFull) return -1;

  uint64_t slots = 1024;
  while (slots < 2ull * (uint64_t)max_voxels) slots <<= 1;
  const uint32_t mask = (uint32_t)(slots - 1);
  int shift = 32;
  for (uint64_t s = slots; s > 1; s >>= 1) --shift;
  uint32_t *cell = (uint32_t *)malloc(slots * sizeof(uint32_t));
  int32_t *vox_of = (int32_t *)malloc(slots * sizeof(int32_t));
  if (!cell || !vox_of) {
    free(cell);
    free(vox_of);
    return -2;
  }
  memset(cell, 0xFF, slots * sizeof(uint32_t));      /* 0xFFFFFFFF = free slot (no cell has that index, checked above) */

  const size_t row = (size_t)T * (size_t)F;
  int64_t nvox = 0;
  for (int64_t i = 0; i < P; ++i) {
    const float *p = points + i * F;
    uint32_t c[3];
    int inside = 1;
    for (int a = 0; a < 3; ++a) {
      const float d = p[a] - range6[a];
      const float q = d / vsize3[a];
      const float fl = floorf(q);
      if (!(fl >= 0.0f) || !(fl < (float)grid_xyz[a])) {
        inside = 0;
        break;
      }
      c[a] = (uint32_t)fl;
    }
    if (!inside) continue;
    const uint32_t key = (c[2] * (uint32_t)gy + c[1]) * (uint32_t)gx + c[0];
    uint32_t s = (key * 0x9E3779B1u) >> shift;
    while (cell[s] != key && cell[s] != 0xFFFFFFFFu) s = (s + 1) & mask;
    int32_t v;
    if (cell[s] == key) {
      v = vox_of[s];
    } else {
      if (nvox == max_voxels) break;      /* the reference loop stops here, it does not skip the point */
      v = (int32_t)nvox++;
      cell[s] = key;
      vox_of[s] = v;
      coords[3 * (size_t)v + 0] = (int32_t)c[2];
      coords[3 * (size_t)v + 1] = (int32_t)c[1];
      coords[3 * (size_t)v + 2] = (int32_t)c[0];
      num_points[v] = 0;
      memset(voxels + (size_t)v * row, 0, row * sizeof(float));
    }
    const int32_t n = num_points[v];
    if (n < T) {
      memcpy(voxels + (size_t)v * row + (size_t)n * F, p, (size_t)F * sizeof(float));
      num_points[v] = n + 1;
    }
  }
  free(cell);
  free(vox_of);
  return nvox;
}
