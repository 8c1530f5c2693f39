/*
 * rslo_host.h -- C ABI of librslo_host.so: the part of the operator boundary that runs in HOST memory.
 *
 * One entry point today: the numpy face of spconv.utils.VoxelGenerator.generate, which the reference calls on the CPU
 * inside forked DataLoader workers (rslo/data/preprocess.py:493 via rslo/builder/voxel_builder.py:36-54,83-94;
 * workers forked at train_hdf5.py:549-553).  A forked child of a process that has initialised HIP cannot use the
 * device, so this face is plain C without any HIP / torch dependency, re-entrant and fork-safe.  Device tensors never
 * take this path: they are voxelized by rslo_voxelize / rslo_plan_encoder (rslo_hip.h).  Same results bit for bit.
 */
#ifndef RSLO_HOST_H_
#define RSLO_HOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RSLO_HOST_API __attribute__((visibility("default")))
#else
#define RSLO_HOST_API
#endif

RSLO_HOST_API int rslo_host_abi_version(void);

/* points [P,F] fp32 in host memory (x,y,z first).  First-come voxel numbering in point order, at most T points per
 * voxel kept in point order, processing stops at the first point that would open voxel number max_voxels, coordinates
 * stored (z,y,x).  Outputs are caller-owned host buffers sized for max_voxels and need not be initialised: rows
 * [0, return value) are written completely.  Returns the voxel count, -1 (bad argument) or -2 (out of memory). */
RSLO_HOST_API int64_t rslo_host_voxelize(const float *points, int64_t P, int F, const float *range6,
                                         const float *vsize3, const int32_t *grid_xyz, int T, int max_voxels,
                                         float *voxels /*[max_voxels,T,F]*/, int32_t *coords /*[max_voxels,3] zyx*/,
                                         int32_t *num_points /*[max_voxels]*/);

#ifdef __cplusplus
}
#endif
#endif /* RSLO_HOST_H_ */
