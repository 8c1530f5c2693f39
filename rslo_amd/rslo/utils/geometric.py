"""Pose algebra on the evaluation side (reference: rslo/utils/geometric.py:376-476).  numpy only; the reference's
`quaternion` package calls (from_float_array / as_rotation_matrix / from_rotation_matrix) are restated in closed
form: q = (w, x, y, z), normalised before conversion like numpy-quaternion does."""
import numpy as np

import rslo.utils.pose_utils_np as pun


def q2m(q):
    """(w,x,y,z) -> 3x3 rotation matrix of the normalised quaternion."""
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def m2q(R):
    """3x3 rotation -> (w,x,y,z), w >= 0 branch-stable (largest of the four squared components first)."""
    R = np.asarray(R, dtype=np.float64)
    t = np.array([1 + R[0, 0] + R[1, 1] + R[2, 2], 1 + R[0, 0] - R[1, 1] - R[2, 2],
                  1 - R[0, 0] + R[1, 1] - R[2, 2], 1 - R[0, 0] - R[1, 1] + R[2, 2]])
    k = int(np.argmax(t))
    s = 2 * np.sqrt(t[k])
    if k == 0:
        q = np.array([s / 4, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif k == 1:
        q = np.array([(R[2, 1] - R[1, 2]) / s, s / 4, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif k == 2:
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, s / 4, (R[1, 2] + R[2, 1]) / s])
    else:
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, s / 4])
    return q if q[0] >= 0 else -q


def expand_rigid_transformation(trans_matrix):
    if trans_matrix.shape == (4, 4):
        return trans_matrix
    if trans_matrix.shape == (3, 4):
        out = np.zeros((4, 4))
        out[3, 3] = 1
        out[:3, :4] = trans_matrix
        return out
    raise ValueError(f"The matrix of shape {trans_matrix.shape} is not allowed!")


def tq_to_RT(tq, expand=False):
    """(t, q wxyz) [7] -> 3x4 (or 4x4) rigid transform."""
    assert tq.shape == (7,)
    RT = np.zeros((3, 4))
    RT[:3, :3] = q2m(tq[3:])
    RT[:3, 3] = tq[:3]
    return expand_rigid_transformation(RT) if expand else RT


def RT_to_tq(RT):
    """3x4 / 4x4 rigid transform -> (t, q wxyz) as a [1,7] row, q on the w >= 0 hemisphere."""
    assert RT.shape in [(3, 4), (4, 4)]
    return np.concatenate([RT[:3, 3].reshape(-1), m2q(RT[:3, :3])]).reshape((-1, 7))


def odom_to_abs_pose(odoms):
    """Frame-to-frame odometries [N,7] -> absolute poses [N,7].  As in the reference (geometric.py:376-406) the first
    pose is the identity and pose i (i >= 1) is odoms[0] o odoms[1] o ... o odoms[i]: odoms[0] seeds the chain."""
    t_prev, r_prev = odoms[0][:3].reshape(1, 3), odoms[0][3:].reshape(1, 4)
    poses = [np.array([0, 0, 0, 1, 0, 0, 0]).reshape(-1, 7)]
    for i in range(1, len(odoms)):
        t_cur, r_cur = odoms[i][:3].reshape(1, 3), odoms[i][3:].reshape(1, 4)
        t_prev, r_prev = t_prev + pun.rotate_vec_by_q(t_cur, r_prev), pun.qmult(r_prev, r_cur)
        poses.append(np.concatenate([t_prev, r_prev], axis=-1))
    return np.concatenate(poses, axis=0)


def gen_voxel_3d_coords(tq_map, pc_range, return_seq=False, format="BHW3"):
    """Cell anchor coordinates of a BEV (or cubic) map (reference: rslo/utils/geometric.py:159-218): x = (j - ox) vx,
    y = (oy - i) vy, z = (k - oz) vz, no half-cell offset -- the same convention as the local<->global maps
    (rslo/data/dataset.py).  tq_map [B,H,W,C] ("BHW3") or [B,C,H,W] ("B3HW") only supplies shape / dtype / device.
    -> [B,H,W,3] / [B,3,H,W] (Z = 1 squeezed), or [B*H*W, 3] with return_seq."""
    from rslo.data.dataset import _cell_centres, _grid_geometry
    assert format in ("BHW3", "B3HW") and tq_map.dim() == 4
    B = tq_map.shape[0]
    H, W = (tq_map.shape[1:3] if format == "BHW3" else tq_map.shape[2:])
    _, vs, origin = _grid_geometry([1, int(H), int(W)], pc_range)
    xyz = _cell_centres(int(W), int(H), 1, origin, vs, tq_map.device, tq_map.dtype)      # [H*W*1, 3], (i, j, k) order
    if return_seq:
        return xyz.repeat(B, 1)
    xyz = xyz.view(1, int(H), int(W), 3).expand(B, int(H), int(W), 3)
    return xyz.permute(0, 3, 1, 2).contiguous() if format == "B3HW" else xyz.contiguous()


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
