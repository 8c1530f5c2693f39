"""Per-cell ("geometric unit") local <-> global pose maps on the BEV grid
(reference: rslo/data/dataset.py:52-208; conventions SURVEY.md App-B 2).

A BEV map has H rows (decreasing y) and W columns (increasing x); cell (i, j) is anchored at
x = (j - ox) vx, y = (oy - i) vy, z = (k - oz) vz with no half-cell offset, where (ox, oy, oz) is the
grid position of the sensor origin.  A cell's local transform (t_l, q) and the global one (t_g, q)
are related by  t_g = R(q)(t_l - x) + x.
"""
import numpy as np
import torch

import rslo.utils.pose_utils as tch_p

REGISTERED_DATASET_CLASSES = {}


def register_dataset(cls, name=None):
    name = cls.__name__ if name is None else name
    assert name not in REGISTERED_DATASET_CLASSES, f"exist class: {REGISTERED_DATASET_CLASSES}"
    REGISTERED_DATASET_CLASSES[name] = cls
    return cls


def get_dataset_class(name):
    assert name in REGISTERED_DATASET_CLASSES, f"available class: {REGISTERED_DATASET_CLASSES}"
    return REGISTERED_DATASET_CLASSES[name]


class Dataset(object):
    """Interface of the sequence readers (rslo/data/dataset.py:32-50)."""
    NumPointFeatures = -1

    def __getitem__(self, index):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    def evaluation(self, prediction, output_dir):
        raise NotImplementedError


def _cell_centres(size_x, size_y, size_z, origin_loc, voxel_size, device, dtype):
    i = torch.arange(size_y, device=device, dtype=dtype).view(-1, 1, 1)
    j = torch.arange(size_x, device=device, dtype=dtype).view(1, -1, 1)
    k = torch.arange(size_z, device=device, dtype=dtype).view(1, 1, -1)
    xv = ((j - float(origin_loc[0])) * float(voxel_size[0])).expand(size_y, size_x, size_z)
    yv = ((-i + float(origin_loc[1])) * float(voxel_size[1])).expand(size_y, size_x, size_z)
    zv = ((k - float(origin_loc[2])) * float(voxel_size[2])).expand(size_y, size_x, size_z)
    return torch.stack([xv, yv, zv], dim=-1).reshape(-1, 3)


def generate_pointwise_local_transformation_tch(tq, spatial_size, origin_loc, voxel_size, inv_trans_factor=-1):
    """Global pose tq [7] -> map [7, (Z,) H, W] of the local transforms every cell would have to
    predict:  t_l = R(q)^-1 (t_g - x) + x."""
    assert inv_trans_factor <= 0, "inverse-range reweighting is not used by the RSLO hot path"
    device, dtype = tq.device, tq.dtype
    if len(spatial_size) == 2:
        size_x, size_y = int(spatial_size[0]), int(spatial_size[1])
        size_z = 1
    elif len(spatial_size) == 3:
        size_x, size_y, size_z = (int(s) for s in spatial_size)
    else:
        raise ValueError()
    xyzv = _cell_centres(size_x, size_y, size_z, origin_loc, voxel_size, device, dtype)
    t_g, q_g = tq[:3], tq[3:]
    qi = tch_p.qinv(q_g[None]).expand(xyzv.shape[0], 4)
    t_l = tch_p.rotate_vec_by_q(t_g[None] - xyzv, qi) + xyzv
    t_map = t_l.reshape(size_y, size_x, size_z, 3)
    q_map = q_g.expand(size_y, size_x, size_z, 4)
    return torch.cat([t_map, q_map], -1).permute(3, 2, 0, 1).squeeze()


def _grid_geometry(spatial_size, pc_range):
    """spatial_size (Z,H,W) -> (grid xyz, cell size xyz, origin location xyz) as python floats."""
    pc = np.asarray(pc_range, dtype=np.float64)
    grid = np.array(list(spatial_size[::-1]), dtype=np.float64)
    vs = (pc[3:] - pc[:3]) / grid
    origin = ((0 - pc[0]) / (pc[3] - pc[0]) * grid[0], (pc[4] - 0) / (pc[4] - pc[1]) * grid[1],
              (0 - pc[2]) / (pc[5] - pc[2]) * grid[2])
    return grid, vs, origin


def from_pointwise_local_transformation_tch(tq_map, pc_range, inv_trans_factor=-1):
    """Local map [B,7,H,W] -> global map [B,7,H,W]:  t_g = R(q)(t_l - x) + x, q normalised."""
    assert inv_trans_factor <= 0
    assert tq_map.dim() == 4
    B, _, H, W = tq_map.shape
    dtype, device = tq_map.dtype, tq_map.device
    _, vs, origin = _grid_geometry([1, H, W], pc_range)
    vs = np.asarray(vs, dtype=np.float32)
    origin = np.asarray(origin, dtype=np.float32)
    xyzv = _cell_centres(W, H, 1, origin, vs, device, dtype)           # [H*W, 3]
    flat = tq_map.permute(0, 2, 3, 1).reshape(-1, 7)
    xyz = xyzv.repeat(B, 1)
    t_g = tch_p.rotate_vec_by_q(flat[:, :3] - xyz, flat[:, 3:]) + xyz
    q_g = torch.nn.functional.normalize(flat[:, 3:].reshape(B, H, W, 4), dim=-1)
    return torch.cat([t_g.reshape(B, H, W, 3), q_g], -1).permute(0, 3, 1, 2).contiguous()


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
