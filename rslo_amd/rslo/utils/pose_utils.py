"""Quaternion algebra of the hot path, (w, x, y, z) convention
(reference: rslo/utils/pose_utils.py:23-181).  Plain torch; shapes [N,4] / [N,3]."""
import torch


def vdot(v1, v2):
    return (v1 * v2).sum(1)


def normalize(x, p=2, dim=0):
    return x / x.norm(p=p, dim=dim).unsqueeze(dim)


def qmult(q1, q2):
    s1, v1 = q1[:, :1], q1[:, 1:]
    s2, v2 = q2[:, :1], q2[:, 1:]
    qs = s1 * s2 - vdot(v1, v2).unsqueeze(1)
    qv = v1 * s2 + v2 * s1 + torch.cross(v1, v2, dim=1)
    return normalize(torch.cat([qs, qv], 1), dim=1)


def qinv(q):
    return torch.cat([q[:, :1], -q[:, 1:]], 1)


def rotate_vec_by_q(t, q):
    """t' = t + 2 qs (qv x t) + 2 qv x (qv x t)."""
    qs, qv = q[:, :1], q[:, 1:]
    b = torch.cross(qv, t, dim=1)
    c = 2 * torch.cross(qv, b, dim=1)
    return t + 2 * b * qs + c


def compose_pose_quaternion(p1, p2):
    q = qmult(p1[:, 3:], p2[:, 3:])
    t = p1[:, :3] + rotate_vec_by_q(p2[:, :3], p1[:, 3:])
    return torch.cat([t, q], 1)


def invert_pose_quaternion(p):
    qi = qinv(p[:, 3:])
    return torch.cat([-rotate_vec_by_q(p[:, :3], qi), qi], 1)


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
