"""A tiny deterministic KITTI-layout store (7 training sequences, a few frames each) for the reader tests: the same
function builds it for the golden generator and for the tests."""
import numpy as np


def build_store(root, write):
    rs = np.random.RandomState(2024)
    for s in range(7):
        n = 4 + s % 3
        pts, nrm, hier, poses, calib = [], [], [], [], []
        Tr = np.eye(4)[:3]
        Tr[:, :3] = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], float)
        Tr[:, 3] = [0.01, -0.07, -0.27]
        T = np.eye(4)
        for i in range(n):
            P = 30 + 3 * i + s
            pts.append(rs.standard_normal((P, 4)).astype(np.float32) * [20, 10, 1, 0.3])
            nn = rs.standard_normal((P, 3)).astype(np.float32)
            nn /= np.linalg.norm(nn, axis=1, keepdims=True)
            nn[::5] = [0, 0, 1]
            nn[1::7] = [0, 0, -1]
            nrm.append(nn)
            hier.append(rs.standard_normal((P // 2, 6)).astype(np.float32))
            yaw = 0.02 * (i + 1) * (1 if s % 2 else -1)
            step = np.eye(4)
            step[:3, :3] = [[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]]
            step[:3, 3] = [0.02 * i, -0.01, 0.9 + 0.05 * s]
            T = T @ step
            poses.append(T[:3].copy())
            calib.append(Tr.copy())
        write(root, "%02d" % s, {"lidar_points": pts, "lidar_normals": nrm, "hier_lidar_points_normals_0.1": hier,
                                 "poses": np.stack(poses), "calib.Tr_velo_to_cam": np.stack(calib)})


QUERIES = [dict(seq_length=2, skip=1, step=1, idx=[0, 1, 5, 12, 20]), dict(seq_length=3, skip=2, step=1, idx=[3, 9, 30]),
           dict(seq_length=2, skip=-1, step=2, idx=[0, 4, 10])]
