"""Collation of per-sample examples into the batched `example` dict of the hot path
(reference: merge_second_batch, rslo/data/preprocess.py:46-134, and example_convert_to_torch,
train_hdf5.py:44-89).  Only the keys the network reads are handled: voxels / num_points (concatenated over the
batch per frame), coordinates (batch index prepended), num_voxels, odometry / icp_odometry, tq_maps.
Works on numpy arrays (DataLoader workers) and on torch tensors (GPU-side voxelization) alike."""
from collections import defaultdict

import numpy as np
import torch


def _cat(xs):
    if isinstance(xs[0], torch.Tensor):
        return torch.cat(xs, dim=0)
    return np.concatenate(xs, axis=0)


def _stack(xs):
    if isinstance(xs[0], torch.Tensor):
        return torch.stack(xs, dim=0)
    return np.stack(xs, axis=0)


def _pad_batch_index(c, b):
    """[M,3] (z,y,x) -> [M,4] (b,z,y,x)."""
    if isinstance(c, torch.Tensor):
        return torch.cat([torch.full((c.shape[0], 1), b, dtype=c.dtype, device=c.device), c], dim=1)
    return np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=b)


def merge_second_batch(batch_list):
    """[batch][key][frame] -> example[key][frame] with the batch merged per frame."""
    merged = defaultdict(list)
    for example in batch_list:
        for k, v in example.items():
            if isinstance(v, list):
                if k not in merged:
                    merged[k] = [[] for _ in range(len(v))]
                for t, vt in enumerate(v):
                    merged[k][t].append(vt)
            else:
                merged[k].append(v)
    ret = {}
    for key, elems in merged.items():
        if key in ("voxels", "num_points"):
            ret[key] = [_cat(e) for e in elems] if isinstance(elems[0], list) else _cat(elems)
        elif key == "coordinates":
            ret[key] = [_cat([_pad_batch_index(c, b) for b, c in enumerate(frame)]) for frame in elems]
        elif key in ("odometry", "icp_odometry"):
            ret[key] = _stack(elems)
        elif key == "metrics":
            ret[key] = elems
        elif key in ("lidar_seqs", "normal_gt_seqs", "hier_points"):
            continue          # raw-point side inputs the configured loss never reads (voxel_odom_net.py:629-654)
        else:                 # num_voxels, tq_maps, ...
            ret[key] = [_stack(e) for e in elems] if isinstance(elems[0], list) else _stack(elems)
    return ret


def example_convert_to_torch(example, dtype=torch.float32, device=None):
    """numpy -> torch on `device` with the reference's dtypes: voxels float, coordinates / num_points int32,
    num_voxels int64 kept on the host (the network reads its shape only)."""
    device = device or torch.device("cuda")
    out = {}
    for k, v in example.items():
        def conv(x, k=k):
            if k == "num_voxels":
                return torch.as_tensor(x, dtype=torch.int64)
            if k in ("coordinates", "num_points"):
                return torch.as_tensor(x, dtype=torch.int32, device=device)
            if isinstance(x, (np.ndarray, torch.Tensor)):
                return torch.as_tensor(x, dtype=dtype, device=device)
            return x
        out[k] = [conv(x) for x in v] if isinstance(v, list) else conv(v)
    return out


# --------------------------------------------------------------------------------------------------------------
# y-flip augmentation (reference: prep_pointcloud, rslo/data/preprocess.py:335-386, flip_odometry :230-245)
# --------------------------------------------------------------------------------------------------------------
def flip_odometry(old_odom):
    """Pose (t, q wxyz) of the scene mirrored in the x-z plane: R' = F R F^T, t' = F t with F = diag(1, -1, 1).
    For a quaternion that conjugation is (w, x, y, z) -> (w, -x, y, -z); the result is put on the w >= 0 hemisphere like
    the reference's matrix round trip does."""
    o = np.asarray(old_odom, dtype=np.float64)
    q = np.array([o[3], -o[4], o[5], -o[6]])
    q = q / np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    return np.concatenate([[o[0], -o[1], o[2]], q], axis=-1)


def flip_points_y(points):
    """In place: y -> -y for the coordinates and for the normal's y component (column 5; column 8 when a second
    normal is present).  Works on numpy arrays and torch tensors (the GPU-resident clouds of the prefetcher)."""
    points[:, 1] = -points[:, 1]
    if points.shape[1] >= 6:
        points[:, 5] = -points[:, 5]
    if points.shape[1] >= 9:
        points[:, 8] = -points[:, 8]
    return points


def random_flip_y(input_dict, points_list, rng=np.random):
    """With probability 1/2 mirror all frames of the sample and every pairwise odometry (and icp_odometry) with it."""
    if not rng.rand() > 0.5:
        return False
    for p in points_list:
        flip_points_y(p)
    for key in ("odometry", "icp_odometry"):
        if input_dict.get(key, None) is not None:
            for i in range(len(input_dict[key])):
                input_dict[key][i] = flip_odometry(input_dict[key][i])
    return True


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
