"""KITTI odometry benchmark metrics on (t, q) pose lists (reference: rslo/utils/kitti_evaluation.py:24-255, itself the
KITTI devkit's evaluate_odometry): relative pose error over sub-sequences of 100..800 m starting every 10 frames,
translation error in m/m and rotation error in rad/m, averaged per segment length / overall / per speed bucket.
Plotting helpers of the reference (matplotlib) are not part of the path."""
import numpy as np

from rslo.utils.geometric import tq_to_RT


class kittiOdomEval:
    def __init__(self):
        self.lengths = [100, 200, 300, 400, 500, 600, 700, 800]
        self.num_lengths = len(self.lengths)
        self.step_size = 10          # one start per second at 10 Hz

    def toCameraCoord(self, pose_mat):
        """LiDAR-frame pose -> camera-frame pose (x right, y down, z forward)."""
        c2l = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
        return np.linalg.inv(c2l) @ pose_mat @ c2l

    def trajectoryDistances(self, poses):
        """Cumulative path length at every frame."""
        xyz = np.stack([p[:3, 3] for p in poses])
        dist = [0]
        for i in range(len(xyz) - 1):
            d = xyz[i] - xyz[i + 1]
            dist.append(dist[i] + np.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2))
        self.distance = dist[-1]
        return dist

    @staticmethod
    def rotationError(pose_error):
        d = 0.5 * (pose_error[0, 0] + pose_error[1, 1] + pose_error[2, 2] - 1.0)
        return np.arccos(max(min(d, 1.0), -1.0))

    @staticmethod
    def translationError(pose_error):
        dx, dy, dz = pose_error[0, 3], pose_error[1, 3], pose_error[2, 3]
        return np.sqrt(dx ** 2 + dy ** 2 + dz ** 2)

    @staticmethod
    def lastFrameFromSegmentLength(dist, first_frame, len_):
        for i in range(first_frame, len(dist)):
            if dist[i] > dist[first_frame] + len_:
                return i
        return -1

    def calcOdomErrors(self, odom_pred, odom_gt):
        """Per-frame odometry error [rad, m] (the last ground-truth entry is skipped, as in the reference)."""
        pred = [tq_to_RT(o, expand=True) for o in odom_pred]
        gt = [tq_to_RT(o, expand=True) for o in odom_gt]
        errs = []
        for i in range(len(gt) - 1):
            e = np.linalg.inv(pred[i]) @ gt[i]
            errs.append([self.rotationError(e), self.translationError(e)])
        return errs

    def calcSequenceErrors(self, poses_result, poses_gt):
        """-> rows [first_frame, r_err / len, t_err / len, len, speed]."""
        gt = [tq_to_RT(p, expand=True) for p in poses_gt]
        res = [tq_to_RT(p, expand=True) for p in poses_result]
        err = []
        self.max_speed = 0
        dist = self.trajectoryDistances(gt)
        for first in range(0, len(gt), self.step_size):
            for len_ in self.lengths:
                last = self.lastFrameFromSegmentLength(dist, first, len_)
                if last == -1 or last >= len(res) or first >= len(res):
                    continue
                d_gt = np.linalg.inv(gt[first]) @ gt[last]
                d_res = np.linalg.inv(res[first]) @ res[last]
                e = np.linalg.inv(d_res) @ d_gt
                speed = len_ / (0.1 * (last - first + 1.0))
                self.max_speed = max(self.max_speed, speed)
                err.append([first, self.rotationError(e) / len_, self.translationError(e) / len_, len_, speed])
        return err

    @staticmethod
    def computeOverallErr(seq_err):
        """-> (average translation error, average rotation error)."""
        n = len(seq_err)
        return sum(e[2] for e in seq_err) / n, sum(e[1] for e in seq_err) / n

    def computeSegmentErr(self, seq_errs, return_seg_err=False):
        per_len = {l: [] for l in self.lengths}
        for e in seq_errs:
            per_len[e[3]].append([e[2], e[1]])
        avg = {l: [np.mean(np.asarray(v)[:, 0]), np.mean(np.asarray(v)[:, 1])] for l, v in per_len.items() if v}
        return (avg, per_len) if return_seg_err else avg

    @staticmethod
    def computeSegmentAvgErr(segment_errs):
        if len(segment_errs) == 0:
            return 0, 0
        n = len(segment_errs)
        return sum(v[0] for v in segment_errs.values()) / n, sum(v[1] for v in segment_errs.values()) / n

    @staticmethod
    def computeSegmentRMSEErr(segment_errs):
        if len(segment_errs) == 0:
            return 0, 0
        n = len(segment_errs)
        return (np.sqrt(sum(v[0] ** 2 for v in segment_errs.values()) / n),
                np.sqrt(sum(v[1] ** 2 for v in segment_errs.values()) / n))

    @staticmethod
    def computeSpeedErr(seq_errs):
        buckets = {s: [] for s in range(2, 25, 2)}
        for e in seq_errs:
            for s in buckets:
                if np.abs(e[4] - s) < 2.0:
                    buckets[s].append([e[2], e[1]])
        return {s: ([np.mean(np.asarray(v)[:, 0]), np.mean(np.asarray(v)[:, 1])] if v else [])
                for s, v in buckets.items()}


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
