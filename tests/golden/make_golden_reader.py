"""Golden vectors for the sequence reader from the REFERENCE'S OWN KittiDatasetHDF5.get_sensor_data / __len__ /
generate_cyc_vo (rslo/data/kitti_dataset_hdf5.py), imported from /root/reference and run over the small deterministic
store of golden_store.py.  h5py is not in this image: `utils.singleton.HDF5File` (the reference's file-handle singleton,
an open(...).read() wrapper) is replaced by one that hands out this repo's numpy GroupStore -- the indexing, clamping,
pose and odometry logic executed is the reference's.   python tests/golden/make_golden_reader.py -> reader_ref.npz"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden_ref as MG          # noqa: E402
from golden_store import QUERIES, build_store   # noqa: E402


def quaternion_standin():
    """numpy-quaternion (not installed) stood in for by scipy's Rotation -- an independent implementation (the same
    stand-in as make_golden_eval.py)."""
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


def main():
    quaternion_standin()
    MG.install_shims()
    # the mirror's store classes, loaded by file path (the reference's `rslo` package is the one on sys.path here)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mirror_reader", os.path.join(ROOT, "rslo_amd/rslo/data/kitti_dataset_hdf5.py"))
    sing = types.ModuleType("utils.singleton")
    store_holder = {}

    class HDF5File:
        def __init__(self, path, *a, **k):
            self.path = path

        def read(self):
            return store_holder["cls"](self.path)
    sing.HDF5File = HDF5File
    pkg = types.ModuleType("utils")
    pkg.__path__ = []
    sys.modules["utils"], sys.modules["utils.singleton"] = pkg, sing
    for name in ("rslo.data.kitti_common", "rslo.utils.visualization"):
        m = types.ModuleType(name)
        m.draw_trajectory = m.pltfig2data = None
        sys.modules[name] = m
    import rslo.data.kitti_dataset_hdf5 as RK            # the reference's module
    mirror = importlib.util.module_from_spec(spec)
    sys.modules["_mirror_reader"] = mirror
    # the mirror module imports `rslo.data.dataset.Dataset/register_dataset` etc. from whatever `rslo` is importable:
    # here that is the reference's package, which has all of them
    import rslo.data.dataset as RD
    RD.REGISTERED_DATASET_CLASSES.pop("KittiDatasetHDF5")     # the mirror class registers under the same name
    spec.loader.exec_module(mirror)
    store_holder["cls"] = mirror.GroupStore
    out = {}
    with tempfile.TemporaryDirectory() as d:
        build_store(d, mirror.write_numpy_store)
        for qi, q in enumerate(QUERIES):
            ds = RK.KittiDatasetHDF5(d, d, seq_length=q["seq_length"], skip=q["skip"], split="train",
                                     num_point_features=7, step=q["step"])
            out["q%d_len" % qi] = np.array(len(ds))
            out["q%d_seq_lens" % qi] = np.array(ds.seq_lens)
            for idx in q["idx"]:
                r = ds.get_sensor_data(idx)
                tag = "q%d_i%d_" % (qi, idx)
                out[tag + "odometry"] = r["odometry"]
                out[tag + "pose_seq"] = np.concatenate(r["pose_seq"], 0)
                out[tag + "seq_idx"] = np.concatenate(r["seq_idx"])
                out[tag + "frame_idx"] = np.concatenate(r["frame_idx"])
                for t, p in enumerate(r["lidar_seq"]):
                    out[tag + "lidar%d" % t] = p
                out[tag + "hier0"] = r["hier_points_seq"][0][0]
    path = os.path.join(HERE, "reader_ref.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), len(out))


if __name__ == "__main__":
    main()
