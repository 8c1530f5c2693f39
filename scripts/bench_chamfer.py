import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle as O
from rslo_amd import capi, synthetic
v, c, n = O.voxelize(synthetic.scan(), synthetic.PC_RANGE, synthetic.VOXEL_SIZE, 10, 40000)
p = O.vfe_mean(v, n)[:, :3][None].copy()
v2, c2, n2 = O.voxelize(synthetic.scan(pose_xy=(0.9, 0.02), scan_seed=1), synthetic.PC_RANGE, synthetic.VOXEL_SIZE, 10, 40000)
q = O.vfe_mean(v2, n2)[:, :3][None].copy()
a, b = torch.from_numpy(q).cuda(), torch.from_numpy(p).cuda()
d, i = capi.chamfer_nn(a, b); torch.cuda.synchronize()
sub = slice(0, a.shape[1], 13)
od, oi = O.chamfer_nn(q[:, sub], p)
print("bit-exact vs oracle on every 13th query:", bool((i[:, sub].cpu().numpy() == oi).all() and (d[:, sub].cpu().numpy() == od).all()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): capi.chamfer_nn(a, b)
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / 20
N, M = a.shape[1], b.shape[1]
print("chamfer N=%d M=%d: %.1f us, %.2f T pair-evals/s, %.1f TFLOP/s (8 flop/pair)" % (N, M, us, N * M / us / 1e6, 8 * N * M / us / 1e6))
