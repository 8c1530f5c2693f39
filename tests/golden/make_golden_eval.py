"""Golden vectors for the evaluation row by IMPORTING the reference's rslo/utils/geometric.py and
rslo/utils/kitti_evaluation.py (authoring container only):

    python tests/golden/make_golden_eval.py   ->  tests/golden/eval_side.npz

Shims: the `quaternion` package (numpy-quaternion, not installed) is stood in for by scipy.spatial.transform.Rotation
-- an independent implementation, not this repository's restatement; matplotlib / transforms3d import-only.
A synthetic 1.6 km drive with noisy predicted odometry is pushed through odom_to_abs_pose, RT_to_tq/tq_to_RT and the
KITTI segment metrics."""
import collections
import collections.abc
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def drive(n=1700, seed=0, noise=0.0):
    """Frame-to-frame odometries (t, q wxyz) of a car on a winding road at ~10 m/s, 10 Hz."""
    r = np.random.default_rng(seed)
    yaw = 0.004 * np.sin(np.arange(n) / 60.0) + 0.002 * np.sin(np.arange(n) / 17.0) + noise * r.normal(size=n) * 0.0005
    pitch = 0.0005 * np.sin(np.arange(n) / 45.0)
    fwd = 1.0 + 0.2 * np.sin(np.arange(n) / 90.0) + noise * r.normal(size=n) * 0.01
    out = np.zeros((n, 7))
    for i in range(n):
        cy, sy, cp, sp = np.cos(yaw[i] / 2), np.sin(yaw[i] / 2), np.cos(pitch[i] / 2), np.sin(pitch[i] / 2)
        q = np.array([cy * cp, -sy * sp, cy * sp, sy * cp])          # yaw (z) then pitch (y)
        out[i] = [fwd[i], 0.01 * np.sin(i / 30.0) + noise * r.normal() * 0.002, 0.0, *(q / np.linalg.norm(q))]
    return out


def main():
    for n in ("Iterable", "Mapping", "Sequence"):
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))
    from scipy.spatial.transform import Rotation as Rot
    quat = types.ModuleType("quaternion")
    quat.from_float_array = lambda a: np.asarray(a, dtype=np.float64)
    quat.as_float_array = lambda q: np.asarray(q, dtype=np.float64)
    quat.as_rotation_matrix = lambda q: Rot.from_quat(np.roll(np.asarray(q, np.float64), -1)).as_matrix()

    def from_rotation_matrix(R):
        x, y, z, w = Rot.from_matrix(np.asarray(R)).as_quat()
        q = np.array([w, x, y, z])
        return q if w >= 0 else -q
    quat.from_rotation_matrix = from_rotation_matrix
    sys.modules["quaternion"] = quat
    for name in ("transforms3d", "transforms3d.quaternions", "transforms3d.euler", "matplotlib",
                 "matplotlib.backends", "matplotlib.backends.backend_pdf", "matplotlib.pyplot", "numba"):
        m = types.ModuleType(name)
        m.switch_backend = lambda *a, **k: None
        m.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
        m.__path__ = []
        sys.modules.setdefault(name, m)
    sys.path.insert(0, os.path.join(REF, "rslo"))
    sys.path.insert(0, REF)
    from rslo.utils import geometric as G
    from rslo.utils.kitti_evaluation import kittiOdomEval

    gt_odo, pred_odo = drive(noise=0.0), drive(noise=1.0)
    gt_abs, pred_abs = G.odom_to_abs_pose(gt_odo), G.odom_to_abs_pose(pred_odo)
    ev = kittiOdomEval()
    seq = np.array(ev.calcSequenceErrors(pred_abs, gt_abs))
    avg_seg = ev.computeSegmentErr(seq.tolist())
    seg = np.array([[k, *v] for k, v in sorted(avg_seg.items())])
    overall = np.array(ev.computeOverallErr(seq.tolist()))
    speed = ev.computeSpeedErr(seq.tolist())
    speed_arr = np.array([[k, *(v if v else [np.nan, np.nan])] for k, v in sorted(speed.items())])
    odo_err = np.array(ev.calcOdomErrors(pred_odo[:200], gt_odo[:200]))
    RT = np.stack([G.tq_to_RT(p) for p in gt_abs[::100]])
    back = np.stack([G.RT_to_tq(m) for m in RT])
    cam = ev.toCameraCoord(G.tq_to_RT(gt_abs[500], expand=True))
    np.savez_compressed(os.path.join(HERE, "eval_side.npz"), gt_odo=gt_odo, pred_odo=pred_odo, gt_abs=gt_abs,
                        pred_abs=pred_abs, seq=seq, seg=seg, overall=overall, speed=speed_arr, odo_err=odo_err, RT=RT,
                        back=back, cam=cam, seg_avg=np.array(ev.computeSegmentAvgErr(avg_seg)),
                        seg_rmse=np.array(ev.computeSegmentRMSEErr(avg_seg)), distance=ev.distance)
    print("segments", len(seq), "overall t_rel %.4f %%  r_rel %.5f deg/m" % (overall[0] * 100, overall[1] * 180 / np.pi),
          "path %.1f m" % ev.distance)


if __name__ == "__main__":
    main()
