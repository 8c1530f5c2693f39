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


def _kitti_like(n, seed):
    g = np.random.default_rng(seed)
    pts = np.concatenate([g.uniform(-75, 75, (n, 1)), g.uniform(-42, 42, (n, 1)), g.uniform(-3.5, 5.5, (n, 1)),
                          g.uniform(0, 1, (n, 4))], 1).astype(np.float32)
    k = len(pts[1::7])
    pts[::7][:k] = pts[1::7]                       # repeated points: several per voxel
    return pts


def test_host_voxelizer_matches_the_oracle():
    """librslo_host.so (the numpy face of VoxelGenerator, plain C, hash-indexed) against the oracle's restatement of
    the classic SECOND loop (dense lookup table): same voxels / coordinates / counts, incl. T truncation, the break at
    max_voxels, an empty cloud and points outside the range."""
    import oracle as O
    from spconv.utils import VoxelGenerator
    rng, vs = [-70.4, -38.4, -3, 70.4, 38.4, 5], [0.1, 0.1, 0.2]
    for n, T, maxv in ((20000, 10, 40000), (20000, 2, 40000), (20000, 10, 1500), (0, 10, 100), (5, 10, 100)):
        pts = _kitti_like(n, n + T)
        if n == 5:
            pts[:, 0] = 500.0                      # nothing inside the range
        vg = VoxelGenerator(vs, rng, T, maxv)
        v, c, m = vg.generate(pts, maxv)
        ov, oc, on = O.voxelize(pts, np.array(rng, np.float32), np.array(vs, np.float32), T, maxv)
        assert v.dtype == np.float32 and c.dtype == np.int32 and m.dtype == np.int32
        assert v.shape == ov.shape and (v == ov).all() and (c == oc).all() and (m == on).all(), (n, T, maxv)
    # hand-made: first-come numbering, (z, y, x) order, slots beyond the count are zero
    pts = np.zeros((4, 7), np.float32)
    pts[:, :3] = [[0.05, 0.05, -2.9], [10.0, 0.05, -2.9], [0.06, 0.06, -2.9], [500, 0, 0]]
    v, c, m = VoxelGenerator(vs, rng, 3, 10).generate(pts)
    assert c.tolist() == [[0, 384, 704], [0, 384, 804]] and m.tolist() == [2, 1]
    assert (v[0, 0] == pts[0]).all() and (v[0, 1] == pts[2]).all() and (v[0, 2] == 0).all() and (v[1, 1:] == 0).all()


def _fork_child(q):
    import rslo_amd  # noqa: F401
    from spconv.utils import VoxelGenerator
    pts = _kitti_like(3000, 1)
    v, c, m = VoxelGenerator([0.1, 0.1, 0.2], [-70.4, -38.4, -3, 70.4, 38.4, 5], 10, 20000).generate(pts, 40000)
    q.put((v.shape, int(c.sum()), int(m.sum())))


def test_numpy_voxel_generator_works_in_forked_workers(monkeypatch):
    """The reference's DataLoader workers are forked (train_hdf5.py:549-553) and call VoxelGenerator.generate on numpy
    arrays (preprocess.py:493).  The numpy face never touches HIP, so it runs in a forked child -- also one whose
    parent has initialised the GPU (simulated: torch.cuda._is_in_bad_fork) -- and gives the parent's result."""
    import multiprocessing as mp
    import torch
    from spconv.utils import VoxelGenerator
    pts = _kitti_like(3000, 1)
    v, c, m = VoxelGenerator([0.1, 0.1, 0.2], [-70.4, -38.4, -3, 70.4, 38.4, 5], 10, 20000).generate(pts, 40000)
    monkeypatch.setattr(torch.cuda, "_is_in_bad_fork", lambda: True)
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fork_child, args=(q,)) for _ in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert res[0] == res[1] == (v.shape, int(c.sum()), int(m.sum()))
    # the voxel_builder surface the data layer uses: a dict of numpy arrays (voxel_builder.py:48-54)
    from rslo.builder import voxel_builder
    from rslo.utils import config_text
    vg = voxel_builder.build(config_text.shipped_config().model.second.voxel_generator)
    out = vg.generate(pts, 40000)
    assert (out["voxels"] == v).all() and (out["coordinates"] == c).all() and (out["num_points_per_voxel"] == m).all()


def test_chamfer_one_direction_functions_stay_cuda_only_like_the_reference():
    """cd.forward / cd.backward take host tensors (librslo_host.so, tests/test_oracle.py); the one-direction functions the
    loss uses are CUDA-only in the reference (chamfer_distance.py:174-175 raises NotImplementedError) and here: no CPU route
    into the training path, and a mixed host / device call is an error, never a silent copy."""
    import pytest
    import torch
    import rslo_amd  # noqa: F401
    from rslo_amd import capi
    from thirdparty.chamfer_distance.chamfer_distance import OneDirectionChamferDistanceWithIdx, cd
    a = torch.zeros(1, 4, 3)
    with pytest.raises((NotImplementedError, capi.RsloHipError)):
        OneDirectionChamferDistanceWithIdx()(a, a)
    d, i = torch.zeros(1, 4), torch.zeros(1, 4, dtype=torch.int32)
    cd.forward(a, a, d, d.clone(), i, i.clone())          # host tensors: served on the host, as in the reference
    assert float(d.abs().max()) == 0.0 and int(i.max()) == 0
