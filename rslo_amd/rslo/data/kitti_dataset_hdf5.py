"""KittiDatasetHDF5: the sequence reader behind the training / evaluation drivers
(reference: rslo/data/kitti_dataset_hdf5.py:28-380).

Group layout it reads (one group per KITTI sequence "00".."21"):
    <seq>/lidar_points[i]                       flat float32 [P_i * 4]    (x, y, z, intensity)
    <seq>/lidar_normals[i]                      flat float32 [P_i * 3]
    <seq>/hier_lidar_points_normals_0.1[i]      flat float32 [Q_i * 6]
    <seq>/poses[i]                              3x4 (or 4x4) camera pose
    <seq>/calib.Tr_velo_to_cam[i]               3x4 (or 4x4)
Storage: an HDF5 file when h5py is importable (the reference's format), or a DIRECTORY in the same layout written by
`write_numpy_store` -- `<dir>/<seq>/<dataset>.npy` for fixed-shape datasets, `<dataset>.values.npy` +
`<dataset>.offsets.npy` for the ragged ones.  This image has no h5py; the numpy store is what the tests and the
synthetic workloads use, and `GroupStore` gives both the same `store[seq][name][i]` access.

What `get_sensor_data(idx)` returns is the reference's dict: the last `seq_length` frames ending at idx with stride
`skip` (clamped to the sequence), each cloud = [P, 4 + 3] points + normals with the (0, 0, +-1) normals zeroed, lidar-frame
poses (t, q) from `cam_pose_to_lidar`, and `odometry` = the relative motions of all frame pairs (i < j), w >= 0.
"""
import os
import random
from pathlib import Path

import numpy as np

from rslo.data.dataset import Dataset, register_dataset
from rslo.utils.geometric import RT_to_tq, expand_rigid_transformation, odom_to_abs_pose
from rslo.utils.kitti_evaluation import kittiOdomEval
from rslo.utils.pose_utils_np import compose_pose_quaternion, invert_pose_quaternion

RAGGED = ("lidar_points", "lidar_normals", "hier_lidar_points_normals_0.1")


def cam_pose_to_lidar(cam_pose, velo_to_cam):
    """Pose in the camera frame -> lidar frame: Tr^-1 @ T @ Tr (rslo/utils/geometric.py:477-490)."""
    T, Tr = expand_rigid_transformation(cam_pose), expand_rigid_transformation(velo_to_cam)
    return np.linalg.inv(Tr) @ T @ Tr


def calc_vo(p0, p1):
    """Relative motion of pose p1 in the frame of p0, [N,7] (rslo/utils/pose_utils_np.py:276-283)."""
    return compose_pose_quaternion(invert_pose_quaternion(p0), p1)


class _Ragged:
    def __init__(self, values, offsets):
        self.values, self.offsets = values, offsets

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += len(self)
        return self.values[self.offsets[i]:self.offsets[i + 1]]


class _NumpyGroup:
    def __init__(self, path):
        self.path = path
        self._cache = {}

    def __getitem__(self, name):
        if name not in self._cache:
            base = os.path.join(self.path, name)
            if os.path.exists(base + ".values.npy"):
                self._cache[name] = _Ragged(np.load(base + ".values.npy", mmap_mode="r"), np.load(base + ".offsets.npy"))
            elif os.path.exists(base + ".npy"):
                self._cache[name] = np.load(base + ".npy", mmap_mode="r")
            else:
                raise KeyError("%s has no dataset %r" % (self.path, name))
        return self._cache[name]


class GroupStore:
    """`store["00"]["lidar_points"][i]` over an HDF5 file (h5py) or a numpy directory store."""

    def __init__(self, path):
        self.path = str(path)
        self._h5 = None
        self._groups = {}
        if not os.path.isdir(self.path):
            try:
                import h5py
            except ImportError as e:
                raise ImportError("%s is an HDF5 file and h5py is not installed; convert it once with "
                                  "rslo.data.kitti_dataset_hdf5.write_numpy_store on a machine that has h5py" % self.path) from e
            self._h5 = h5py.File(self.path, "r", libver="latest", swmr=True)

    def __getitem__(self, seq):
        if self._h5 is not None:
            return self._h5[seq]
        if seq not in self._groups:
            p = os.path.join(self.path, seq)
            if not os.path.isdir(p):
                raise KeyError("no sequence %r in %s" % (seq, self.path))
            self._groups[seq] = _NumpyGroup(p)
        return self._groups[seq]


def write_numpy_store(root, seq, datasets):
    """datasets: {name: list of per-frame arrays (ragged names) or an [N, ...] array} -> <root>/<seq>/..."""
    d = os.path.join(str(root), seq)
    os.makedirs(d, exist_ok=True)
    for name, val in datasets.items():
        if name in RAGGED or isinstance(val, (list, tuple)):
            flat = [np.asarray(v, np.float32).reshape(-1) for v in val]
            np.save(os.path.join(d, name + ".values.npy"), np.concatenate(flat) if flat else np.zeros((0,), np.float32))
            np.save(os.path.join(d, name + ".offsets.npy"), np.cumsum([0] + [len(f) for f in flat]).astype(np.int64))
        else:
            np.save(os.path.join(d, name + ".npy"), np.asarray(val))


@register_dataset
class KittiDatasetHDF5(Dataset):
    def __init__(self, root_path, info_path, seq_length=2, skip=1, random_skip=False, split="train", prep_func=None,
                 cycle_constraint=False, num_point_features=None, step=1):
        assert info_path is not None
        assert split in ["train", "val", "eval_train", "test", "train+test"]
        assert skip != 0 and step > 0
        self._root_path = Path(root_path)
        self.odom_eval = kittiOdomEval()
        self._prep_func = prep_func
        self._local_seq_length = seq_length
        self._skip = self._actual_skip = skip
        self._random_skip = random_skip
        self.seq_segments = None
        self._step = step
        self.NumPointFeatures = num_point_features
        self.file_path = info_path
        if split == "train":
            self.seqs = ["%02d" % i for i in range(7)]
        elif split == "val":
            self.seqs = ["%02d" % i for i in range(7, 11)]
        elif split == "eval_train":
            self.seqs = ["00"]
        elif split == "train+test":
            self.seqs = ["%02d" % i for i in range(7)] + ["%02d" % i for i in range(11, 22)]
        else:
            raise ValueError("split 'test' has no sequence list in the reference either (kitti_dataset_hdf5.py:62-70)")
        self._store = None
        self.info_preprocess(self.file_path, self.seqs)

    def store(self):
        if self._store is None:          # opened lazily: one handle per DataLoader worker process
            self._store = GroupStore(self.file_path)
        return self._store

    def info_preprocess(self, file_path, seqs):
        f = GroupStore(file_path)
        self.seq_lens = [len(f[s]["lidar_points"]) for s in seqs]

    def __len__(self):
        return int((np.sum(self.seq_lens) + self._step - 1) // self._step)

    def __getitem__(self, idx):
        idx, seed = idx if isinstance(idx, (tuple, list)) else (idx, None)
        input_dict = self.get_sensor_data(idx)
        if self._prep_func is None:
            return input_dict
        example = self._prep_func(input_dict=input_dict, seed=seed)
        example["velodyne_path"] = input_dict["velodyne_path"]
        return example

    def _locate(self, idx):
        cum = np.insert(np.cumsum(np.array(self.seq_lens)), 0, 0)
        seq_idx = int(np.nonzero(cum > idx)[0][0] - 1)
        return seq_idx, int(idx - cum[seq_idx])

    def _get_info_from_idx(self, idx):
        return int(self.seqs[self._locate(self._step * idx)[0]])

    def get_sensor_data(self, query, with_lidar=True):
        seq_idx, frame_idx = self._locate(self._step * query)
        abs_seq_idx = int(self.seqs[seq_idx])
        info = self.store()["%02d" % abs_seq_idx]
        if self._random_skip:
            choices = (list(np.arange(1, self._skip + 1)) if self._skip > 0 else
                       list(np.arange(self._skip, 0)) + list(np.arange(1, -self._skip + 1)))
            skip = random.choice(choices)
        else:
            skip = self._skip
        n = len(info["lidar_points"])
        local_seq = np.array([frame_idx - i for i in range(skip * (self._local_seq_length - 1), -np.sign(skip), -skip)], dtype=int)
        local_seq = np.clip(local_seq, 0, n - 1)
        res = {"skip": [skip] * self._local_seq_length, "seq_idx": [], "frame_idx": [], "lidar_seq": [], "pose_seq": [],
               "odometry": None, "calib/Tr_velo_to_cam": [], "velodyne_path": [], "hier_points_seq": [], "icp_pose_seq": []}
        for i in local_seq:
            if i == 4148 and abs_seq_idx == 19:      # a frame the reference skips (kitti_dataset_hdf5.py:230-233)
                i = i + 1
            if with_lidar:
                pts = np.asarray(info["lidar_points"][i]).reshape([-1, 4])
                if self.NumPointFeatures == 6:
                    pts = pts[:, :3]
                normals = np.asarray(info["lidar_normals"][i]).reshape([-1, 3])
                points = np.concatenate([pts, normals], axis=-1)
                tail = points[:, -3:]
                tail[np.abs(tail) == [0, 0, 1]] = 0          # component-wise, as the reference's boolean index does
                hier = [np.asarray(info["hier_lidar_points_normals_0.1"][i]).reshape([-1, 6])]
            else:
                points, hier = None, None
            res["lidar_seq"].append(points)
            res["hier_points_seq"].append(hier)
            res["pose_seq"].append(RT_to_tq(cam_pose_to_lidar(np.asarray(info["poses"][i]),
                                                              np.asarray(info["calib.Tr_velo_to_cam"][i]))))
            res["seq_idx"].append(np.array([abs_seq_idx]))
            res["frame_idx"].append(np.array([frame_idx]))
            res["calib/Tr_velo_to_cam"].append(np.asarray(info["calib.Tr_velo_to_cam"][i]))
        res["odometry"] = self.generate_cyc_vo(res["pose_seq"])
        return res

    def generate_cyc_vo(self, pose_seq):
        assert len(pose_seq) > 1
        vos = []
        for i in range(len(pose_seq)):
            for j in range(i + 1, len(pose_seq)):
                vo = calc_vo(pose_seq[i], pose_seq[j]).squeeze(axis=0)
                vo[3:] *= np.sign(vo[3])
                vos.append(vo)
        return np.stack(vos, axis=0)

    def evaluation(self, prediction, output_dir=None):
        """KITTI odometry errors of predicted relative motions against the store's ground truth (no plotting)."""
        gts = np.stack([self.get_sensor_data(i, with_lidar=False)["odometry"].reshape(-1) for i in range(len(self))], 0)
        preds = np.concatenate([np.concatenate([np.asarray(p["translation_preds"].detach().cpu()),
                                                np.asarray(p["rotation_preds"].detach().cpu())], -1)
                                for p in prediction], 0)
        errors = self.odom_eval.calcSequenceErrors(odom_to_abs_pose(preds), odom_to_abs_pose(gts))
        seg = self.odom_eval.computeSegmentErr(errors)
        return {"error": {"kitti_error": seg, "kitti_avg_error": self.odom_eval.computeSegmentAvgErr(seg)}, "plot": {}}
