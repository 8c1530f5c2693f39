/*
 * rslo_host.h -- C ABI of librslo_host.so: the part of the operator boundary that runs in HOST memory.
 *
 * Two groups of entry points: the HOST functions of thirdparty.chamfer_distance (below), and the numpy face of
 * spconv.utils.VoxelGenerator.generate, which the reference calls on the CPU
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

/* thirdparty.chamfer_distance `cd.forward` / `cd.backward`: the reference computes these on the host when the tensors are not
 * CUDA tensors (thirdparty/chamfer_distance/chamfer_distance.cpp:147-190 and :192-234, chosen at chamfer_distance.py:34,61).
 * Exhaustive nearest neighbour in both directions: squared distance ((dx dx + dy dy) + dz dz) in fp32 without contraction,
 * strict '<' scanning upwards (lowest index wins ties) -- dist and idx equal the reference's and rslo_chamfer_nn's bit for
 * bit; backward zeroes both gradients and adds the two directions in the reference's order.  Return 0, -1 on bad arguments. */
RSLO_HOST_API int rslo_host_chamfer_forward(const float *xyz1 /*[b,n,3]*/, const float *xyz2 /*[b,m,3]*/, int b, int n, int m,
                                            float *dist1 /*[b,n]*/, float *dist2 /*[b,m]*/, int32_t *idx1 /*[b,n]*/,
                                            int32_t *idx2 /*[b,m]*/);
RSLO_HOST_API int rslo_host_chamfer_backward(const float *xyz1, const float *xyz2, int b, int n, int m, const float *graddist1,
                                             const float *graddist2, const int32_t *idx1, const int32_t *idx2,
                                             float *gradxyz1 /*[b,n,3]*/, float *gradxyz2 /*[b,m,3]*/);

#ifdef __cplusplus
}
#endif
#endif /* RSLO_HOST_H_ */
