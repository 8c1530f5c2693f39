"""ctypes binding of librslo_host.so (include/rslo_host.h): the host-memory face of the operator boundary.

Only numpy arrays come here (spconv.utils.VoxelGenerator.generate in DataLoader workers, the way the reference calls
it: rslo/data/preprocess.py:493).  No HIP, no torch: importing and calling this module is safe in a forked child of a
process that owns a GPU context.  The library is built by rslo_amd.build.build_host(); a missing library raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librslo_host.so")
_lib = None

SIGNATURES = {
    "rslo_host_abi_version": (C.c_int, []),
    "rslo_host_voxelize": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("librslo_host.so not found at %s -- build it with `python -m rslo_amd.build`" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def voxelize(points, pc_range, voxel_size, grid_xyz, max_points, max_voxels):
    """points [P,F] numpy -> (voxels [M,T,F] f32, coords [M,3] i32 zyx, num_points [M] i32), M <= max_voxels."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    if pts.ndim != 2:
        raise ValueError("points must be [P, F]")
    P, F = pts.shape
    r = np.ascontiguousarray(pc_range, dtype=np.float32)
    v = np.ascontiguousarray(voxel_size, dtype=np.float32)
    g = np.ascontiguousarray(grid_xyz, dtype=np.int32)
    vox = np.empty((max_voxels, max_points, F), np.float32)
    coords = np.empty((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    n = lib().rslo_host_voxelize(pts.ctypes.data, P, F, r.ctypes.data, v.ctypes.data, g.ctypes.data, int(max_points),
                                 int(max_voxels), vox.ctypes.data, coords.ctypes.data, num.ctypes.data)
    if n < 0:
        raise RuntimeError("rslo_host_voxelize failed (%d)" % n)
    return vox[:n], coords[:n], num[:n]
