"""voxel_builder.build (reference: rslo/builder/voxel_builder.py:36-95): same config mutations
(block_filtering forced on, block_factor >= 1, block_size default 8, height_threshold default 0.2 only
when 0), same `_VoxelGenerator` surface (dict-returning generate, grid_size property)."""
import numpy as np
from spconv.utils import VoxelGenerator


class _VoxelGenerator(VoxelGenerator):
    @property
    def grid_size(self):
        r = np.array(self.point_cloud_range)
        g = (r[3:] - r[:3]) / np.array(self.voxel_size)
        return np.round(g).astype(np.int64)

    def generate(self, points, max_voxels=None):
        res = super().generate(points, max_voxels)
        return {"voxels": res[0], "coordinates": res[1], "num_points_per_voxel": res[2]}


def build(voxel_config):
    voxel_config.block_filtering = True
    voxel_config.block_factor = max(1, voxel_config.block_factor)
    voxel_config.block_size = voxel_config.block_size if voxel_config.block_size > 0 else 8
    voxel_config.height_threshold = voxel_config.height_threshold if voxel_config.height_threshold != 0 else 0.2
    return _VoxelGenerator(
        voxel_size=list(voxel_config.voxel_size), point_cloud_range=list(voxel_config.point_cloud_range),
        max_num_points=voxel_config.max_number_of_points_per_voxel, max_voxels=20000, full_mean=False,
        block_filtering=voxel_config.block_filtering, block_factor=voxel_config.block_factor,
        block_size=voxel_config.block_size, height_threshold=voxel_config.height_threshold)
