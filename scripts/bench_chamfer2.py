"""Chamfer search timings at the bench size: exhaustive scan vs pruned search, aligned and misaligned clouds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle as O, rslo_amd
from rslo_amd import capi, synthetic as S
v, c, n = O.voxelize(S.scan(), S.PC_RANGE, S.VOXEL_SIZE, 10, 40000)
p = O.vfe_mean(v, n)[:, :3].copy(); N = len(p)
th = np.float32(2.2)
Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
Rx = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]], np.float32)
sets = {"aligned (0.8 m shift)": np.stack([p + np.float32([0.8, 0.05, 0.0])] * 4),
        "yaw 126 deg": np.stack([p @ Rz.T] * 4), "arbitrary rotation + 5 m": np.stack([p @ (Rx @ Rz).T + np.float32(5.0)] * 4)}
a = torch.from_numpy(np.stack([p] * 4)).cuda()
for name, tgt in sets.items():
    t = torch.from_numpy(tgt.astype(np.float32)).cuda()
    for m in ("brute", "grid"):
        for _ in range(2): capi.chamfer_nn(a, t, method=m)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): capi.chamfer_nn(a, t, method=m)
        e1.record(); torch.cuda.synchronize()
        print("%-28s %-6s 4 x %d x %d : %8.1f us" % (name, m, N, N, 1e3 * e0.elapsed_time(e1) / 10), flush=True)
