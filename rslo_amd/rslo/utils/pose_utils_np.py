"""numpy quaternion/pose helpers used by evaluation and result gathering (reference: rslo/utils/pose_utils_np.py:114-275).
Quaternions are (w, x, y, z), poses (t[3], q[4])."""
import numpy as np


def vdot(v1, v2):
    return np.sum(v1 * v2, axis=1, keepdims=True)


def normalize(x, p=2, dim=0, eps=1e-6):
    """x / (|x|_p + eps) along `dim` -- note the eps INSIDE the division (reference :127-141): a unit quaternion
    comes back shorter by 1e-6."""
    return x / (np.linalg.norm(x, ord=p, axis=dim, keepdims=True) + eps)


def qmult(q1, q2):
    """Hamilton product of [N,4] quaternions, re-normalised with eps = 1e-6 like the reference (:144-163).
    (The reference's vdot drops the column dimension, which only broadcasts correctly for N = 1 -- the way the
    evaluation calls it; here the product is right for every N.)"""
    s1, v1 = q1[:, :1], q1[:, 1:]
    s2, v2 = q2[:, :1], q2[:, 1:]
    q = np.concatenate((s1 * s2 - vdot(v1, v2), v1 * s2 + v2 * s1 + np.cross(v1, v2, axis=1)), axis=1)
    return normalize(q, dim=1)


def qinv(q):
    return np.concatenate((q[:, :1], -q[:, 1:]), axis=1)


def rotate_vec_by_q(t, q):
    """t' = t + 2 qs (qv x t) + 2 qv x (qv x t)."""
    qs, qv = q[:, :1], q[:, 1:]
    b = np.cross(qv, t, axis=1)
    return t + 2 * b * qs + 2 * np.cross(qv, b, axis=1)


def compose_pose_quaternion(p1, p2):
    return np.concatenate((p1[:, :3] + rotate_vec_by_q(p2[:, :3], p1[:, 3:]), qmult(p1[:, 3:], p2[:, 3:])), axis=1)


def invert_pose_quaternion(p):
    qi = qinv(p[:, 3:])
    return np.concatenate((-rotate_vec_by_q(p[:, :3], qi), qi), axis=1)


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
