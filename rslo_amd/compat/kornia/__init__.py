"""Stand-in for the two kornia==0.4.0 functions the RSLO hot path calls (freeze.yml:212; call sites
rslo/core/losses.py:234,244,359, rslo/models/voxel_odom_net.py:675,729, rslo/models/odom_pred.py:291).

kornia is a third-party dependency that is NOT part of the reference tree and is not installed here, so
this restates the published 0.4.0 algorithm: quaternions are (x, y, z, w); quaternion -> matrix
L2-normalises its input first (eps 1e-12); matrix -> quaternion uses the four trace branches with
eps = 1e-8.  Parity is pinned by algebraic known-answer tests (tests/test_golden_host.py), not by kornia.
Only used when a real `kornia` is not importable.
"""
import torch
import torch.nn.functional as F


def normalize_quaternion(quaternion, eps=1e-12):
    return F.normalize(quaternion, p=2, dim=-1, eps=eps)


def quaternion_to_rotation_matrix(quaternion):
    if quaternion.shape[-1] != 4:
        raise ValueError("Input must be a tensor of shape (*, 4). Got {}".format(quaternion.shape))
    q = normalize_quaternion(quaternion)
    x, y, z, w = torch.chunk(q, chunks=4, dim=-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    m = torch.stack([1.0 - (tyy + tzz), txy - twz, txz + twy,
                     txy + twz, 1.0 - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, 1.0 - (txx + tyy)], dim=-1).view(-1, 3, 3)
    if len(quaternion.shape) == 1:
        m = torch.squeeze(m, dim=0)
    return m


def rotation_matrix_to_quaternion(rotation_matrix, eps=1e-8):
    if rotation_matrix.shape[-2:] != (3, 3):
        raise ValueError("Input size must be a (*, 3, 3) tensor. Got {}".format(rotation_matrix.shape))

    def safe_div(num, den):
        tiny = torch.finfo(num.dtype).tiny
        return num / torch.clamp(den, min=tiny)

    v = rotation_matrix.reshape(*rotation_matrix.shape[:-2], 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(v, chunks=9, dim=-1)
    trace = m00 + m11 + m22

    def trace_pos():
        sq = torch.sqrt(trace + 1.0) * 2.0
        return torch.cat([safe_div(m21 - m12, sq), safe_div(m02 - m20, sq), safe_div(m10 - m01, sq), 0.25 * sq], -1)

    def cond_1():
        sq = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0
        return torch.cat([0.25 * sq, safe_div(m01 + m10, sq), safe_div(m02 + m20, sq), safe_div(m21 - m12, sq)], -1)

    def cond_2():
        sq = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0
        return torch.cat([safe_div(m01 + m10, sq), 0.25 * sq, safe_div(m12 + m21, sq), safe_div(m02 - m20, sq)], -1)

    def cond_3():
        sq = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0
        return torch.cat([safe_div(m02 + m20, sq), safe_div(m12 + m21, sq), 0.25 * sq, safe_div(m10 - m01, sq)], -1)

    where_2 = torch.where(m11 > m22, cond_2(), cond_3())
    where_1 = torch.where((m00 > m11) & (m00 > m22), cond_1(), where_2)
    return torch.where(trace > 0.0, trace_pos(), where_1)
