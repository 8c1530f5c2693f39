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
    "rslo_host_chamfer_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "rslo_host_chamfer_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
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


def _host_f32(t, name):
    import torch
    if t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("%s must be a contiguous float32 host tensor" % name)
    return t.data_ptr()


def _host_i32(t, name):
    import torch
    if t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
        raise ValueError("%s must be a contiguous int32 host tensor" % name)
    return t.data_ptr()


def chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """cd.forward on host tensors (caller-allocated outputs, like the reference's pybind entry point)."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    rc = lib().rslo_host_chamfer_forward(_host_f32(xyz1, "xyz1"), _host_f32(xyz2, "xyz2"), b, n, m, _host_f32(dist1, "dist1"),
                                         _host_f32(dist2, "dist2"), _host_i32(idx1, "idx1"), _host_i32(idx2, "idx2"))
    if rc:
        raise RuntimeError("rslo_host_chamfer_forward: bad arguments (empty cloud?)")


def chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """cd.backward on host tensors: gradxyz1 / gradxyz2 are overwritten."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    rc = lib().rslo_host_chamfer_backward(_host_f32(xyz1, "xyz1"), _host_f32(xyz2, "xyz2"), b, n, m,
                                          _host_f32(graddist1, "graddist1"), _host_f32(graddist2, "graddist2"),
                                          _host_i32(idx1, "idx1"), _host_i32(idx2, "idx2"), _host_f32(gradxyz1, "gradxyz1"),
                                          _host_f32(gradxyz2, "gradxyz2"))
    if rc:
        raise RuntimeError("rslo_host_chamfer_backward: bad arguments")
