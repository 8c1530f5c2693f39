"""spconv.utils.VoxelGenerator on the MI355X voxelizer (rslo_voxelize).

Mirrors the constructor and `generate` of the spconv_plus class the reference builds in
rslo/builder/voxel_builder.py:83-94 (extra block-filtering kwargs accepted; with the shipped
height_threshold = -1 the filter keeps everything, which is the only mode implemented).

`generate(points, max_voxels)` takes
  * a CUDA tensor [P,F]  -> voxelized by rslo_voxelize on the GPU, returns CUDA tensors (the training path:
                            rslo_amd.workload.ExamplePrefetcher / rslo_plan_encoder keep the clouds resident), or
  * a numpy array [P,F]  -> voxelized in HOST memory by librslo_host.so (include/rslo_host.h), returns numpy arrays.

Where an input lives decides where it is voxelized; nothing is moved behind the caller's back.  The numpy face is what
the reference's DataLoader workers call (rslo/data/preprocess.py:493): they are forked (train_hdf5.py:549-553), and a
forked child of a process that has initialised HIP cannot use the device -- so that face is plain C without HIP /
torch, re-entrant and fork-safe, and train_hdf5.py's loaders run unchanged.  Both faces give the same bits
(tests/test_gpu_kernels.py::test_host_voxelizer_equals_device_voxelizer).
"""
import numpy as np
import torch

from rslo_amd import capi


class VoxelGenerator:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, full_mean=False,
                 block_filtering=False, block_factor=1, block_size=8, height_threshold=-1.0, **kwargs):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        self._grid_size = np.round(grid_size).astype(np.int64)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = int(max_num_points)
        self._max_voxels = int(max_voxels)
        if full_mean:
            raise NotImplementedError("full_mean=True is not used by the RSLO path (voxel_builder.py:88)")
        if block_filtering and height_threshold > 0:
            raise NotImplementedError("block filtering with a positive height_threshold is outside the hot path "
                                      "(shipped config: height_threshold = -1, kitti_train_ours.prototxt:17)")
        self._block = (block_filtering, block_factor, block_size, height_threshold)

    def generate(self, points, max_voxels=None):
        maxv = int(max_voxels or self._max_voxels)
        if isinstance(points, np.ndarray):
            from rslo_amd import hostlib
            return hostlib.voxelize(points, self._point_cloud_range, self._voxel_size, self._grid_size,
                                    self._max_num_points, maxv)
        pts = points.contiguous().float()
        vox, coords, num, nvox = capi.voxelize(pts, self._point_cloud_range, self._voxel_size, self._grid_size,
                                               self._max_num_points, maxv)
        M = int(nvox.item())
        return vox[:M], coords[:M], num[:M]

    def generate_many(self, clouds, max_voxels=None):
        """Voxelize several CUDA clouds with ONE host read of the voxel counts (generate() syncs per cloud).
        Returns a list of (voxels, coordinates, num_points_per_voxel)."""
        maxv = int(max_voxels or self._max_voxels)
        outs = [capi.voxelize(p.contiguous().float(), self._point_cloud_range, self._voxel_size, self._grid_size,
                              self._max_num_points, maxv) for p in clouds]
        counts = torch.cat([o[3] for o in outs]).tolist()
        return [(o[0][:m], o[1][:m], o[2][:m]) for o, m in zip(outs, counts)]

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size
