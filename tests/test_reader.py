"""Sequence reader against vectors from the reference's own KittiDatasetHDF5 run over the same small
store (tests/golden/make_golden_reader.py): dataset length, frame selection with stride / clamping / step, lidar-frame
poses, pairwise odometry, assembled point rows (normals (0,0,+-1) zeroed), hierarchical points."""
import os
import sys

import numpy as np
import pytest

import rslo_amd  # noqa: F401

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from golden_store import QUERIES, build_store   # noqa: E402


@pytest.fixture(scope="module")
def store(tmp_path_factory):
    from rslo.data.kitti_dataset_hdf5 import write_numpy_store
    d = str(tmp_path_factory.mktemp("kitti_store"))
    build_store(d, write_numpy_store)
    return d


def test_reader_matches_reference_vectors(store):
    from rslo.data.dataset import get_dataset_class
    g = np.load(os.path.join(GOLD, "reader_ref.npz"))
    cls = get_dataset_class("KittiDatasetHDF5")
    for qi, q in enumerate(QUERIES):
        ds = cls(store, store, seq_length=q["seq_length"], skip=q["skip"], split="train", num_point_features=7,
                 step=q["step"])
        assert len(ds) == int(g["q%d_len" % qi]) and list(ds.seq_lens) == list(g["q%d_seq_lens" % qi])
        for idx in q["idx"]:
            r = ds.get_sensor_data(idx)
            tag = "q%d_i%d_" % (qi, idx)
            np.testing.assert_allclose(r["odometry"], g[tag + "odometry"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(np.concatenate(r["pose_seq"], 0), g[tag + "pose_seq"], rtol=1e-12, atol=1e-12)
            assert (np.concatenate(r["seq_idx"]) == g[tag + "seq_idx"]).all()
            assert (np.concatenate(r["frame_idx"]) == g[tag + "frame_idx"]).all()
            assert len(r["lidar_seq"]) == q["seq_length"]
            for t, p in enumerate(r["lidar_seq"]):
                assert p.shape[1] == 7 and (p == g[tag + "lidar%d" % t]).all()
            assert (r["hier_points_seq"][0][0] == g[tag + "hier0"]).all()
            assert (r["odometry"][:, 3] >= 0).all()


def test_reader_feeds_the_collate_and_missing_h5py_is_explained(store, tmp_path):
    from rslo.data.kitti_dataset_hdf5 import GroupStore, KittiDatasetHDF5
    ds = KittiDatasetHDF5(store, store, seq_length=2, skip=1, split="eval_train", num_point_features=7)
    item = ds[(1, 123)]                 # (index, seed) as the samplers yield it; no prep_func -> the raw dict
    assert len(item["lidar_seq"]) == 2 and item["odometry"].shape == (1, 7)
    f = tmp_path / "kitti.h5"
    f.write_bytes(b"not really hdf5")
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="write_numpy_store"):
            GroupStore(str(f))
