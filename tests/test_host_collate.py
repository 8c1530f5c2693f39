"""a2: merge_second_batch coordinate padding / concatenation (reference rslo/data/preprocess.py:46-134)."""
import numpy as np
import torch

import rslo_amd  # noqa: F401
from rslo.data.preprocess import example_convert_to_torch, merge_second_batch


def sample(seed, n0, n1):
    r = np.random.default_rng(seed)
    return {"voxels": [r.random((n0, 10, 7), np.float32), r.random((n1, 10, 7), np.float32)],
            "num_points": [r.integers(1, 10, n0).astype(np.int32), r.integers(1, 10, n1).astype(np.int32)],
            "coordinates": [r.integers(0, 40, (n0, 3)).astype(np.int32), r.integers(0, 40, (n1, 3)).astype(np.int32)],
            "num_voxels": [np.array([n0], np.int64), np.array([n1], np.int64)],
            "icp_odometry": np.zeros((1, 7), np.float32),
            "lidar_seqs": [r.random((50, 4)), r.random((60, 4))]}


def test_merge_second_batch_layout():
    a, b = sample(0, 5, 7), sample(1, 3, 4)
    m = merge_second_batch([a, b])
    assert [v.shape for v in m["voxels"]] == [(8, 10, 7), (11, 10, 7)]
    assert [v.shape for v in m["num_points"]] == [(8,), (11,)]
    c0 = m["coordinates"][0]
    assert c0.shape == (8, 4) and c0[:5, 0].tolist() == [0] * 5 and c0[5:, 0].tolist() == [1] * 3
    assert (c0[:5, 1:] == a["coordinates"][0]).all() and (c0[5:, 1:] == b["coordinates"][0]).all()
    assert m["num_voxels"][0].shape == (2, 1) and m["num_voxels"][1].reshape(-1).tolist() == [7, 4]
    assert m["icp_odometry"].shape == (2, 1, 7)
    assert "lidar_seqs" not in m
    t = example_convert_to_torch(m, device=torch.device("cpu"))
    assert t["coordinates"][0].dtype == torch.int32 and t["voxels"][1].dtype == torch.float32
    assert t["num_voxels"][0].dtype == torch.int64 and t["num_voxels"][0].shape[0] == 2


def test_merge_accepts_torch_inputs():
    a, b = sample(2, 4, 4), sample(3, 2, 6)
    ta = {k: ([torch.from_numpy(x) for x in v] if isinstance(v, list) else torch.from_numpy(v)) for k, v in a.items()}
    tb = {k: ([torch.from_numpy(x) for x in v] if isinstance(v, list) else torch.from_numpy(v)) for k, v in b.items()}
    m, n = merge_second_batch([ta, tb]), merge_second_batch([a, b])
    assert (m["coordinates"][1].numpy() == n["coordinates"][1]).all()
    assert (m["voxels"][0].numpy() == n["voxels"][0]).all()


def test_flip_y_augmentation_matches_matrix_conjugation():
    """flip_odometry == the reference's route through rotation matrices (F R F^T, F t), checked with scipy."""
    from scipy.spatial.transform import Rotation as Rot
    from rslo.data.preprocess import flip_odometry, flip_points_y, random_flip_y
    r = np.random.default_rng(0)
    F = np.diag([1.0, -1.0, 1.0])
    for _ in range(20):
        q = r.normal(size=4); q /= np.linalg.norm(q)
        t = r.normal(size=3)
        new = flip_odometry(np.concatenate([t, q]))
        R_old = Rot.from_quat(np.roll(q, -1)).as_matrix()
        x, y, z, w = Rot.from_matrix(F @ R_old @ F.T).as_quat()
        q_ref = np.array([w, x, y, z]) * (1 if w >= 0 else -1)
        np.testing.assert_allclose(new[:3], F @ t, atol=1e-12)
        np.testing.assert_allclose(new[3:], q_ref, atol=1e-9)
    pts = r.normal(size=(10, 7)).astype(np.float32)
    ref = pts.copy()
    flip_points_y(pts)
    assert (pts[:, 1] == -ref[:, 1]).all() and (pts[:, 5] == -ref[:, 5]).all() and (pts[:, [0, 2, 3, 4, 6]] == ref[:, [0, 2, 3, 4, 6]]).all()
    t = torch.from_numpy(ref.copy())
    assert torch.equal(flip_points_y(t), torch.from_numpy(pts))

    class Always:
        def rand(self):
            return 0.9
    d = {"odometry": [np.array([1.0, 2.0, 3.0, 1.0, 0.0, 0.0, 0.0])], "icp_odometry": None}
    assert random_flip_y(d, [ref.copy()], rng=Always()) and d["odometry"][0][1] == -2.0


def test_numpy_voxel_generator_refuses_in_a_forked_gpu_child(monkeypatch):
    """spconv.utils.VoxelGenerator.generate(numpy) inside a forked child of a GPU process: clear error, no hang."""
    import pytest
    import torch
    from spconv.utils import VoxelGenerator
    vg = VoxelGenerator([0.1, 0.1, 0.2], [-70.4, -38.4, -3, 70.4, 38.4, 5], 10, 20000)
    monkeypatch.setattr(torch.cuda, "_is_in_bad_fork", lambda: True)
    with pytest.raises(RuntimeError, match="spawn"):
        vg.generate(np.zeros((10, 7), np.float32))
