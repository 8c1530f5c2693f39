"""Nearest-neighbour search at the step's shape: 4 frame pairs of the synthetic 64-ring scans (ragged, ~115k points
each), both directions, pruned search vs the exhaustive scan (once, for the equality check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rslo_amd
from rslo_amd import capi, synthetic as S
pairs = [S.frame_pair(i)[:2] for i in range(4)]
if os.environ.get("RAW", "0") != "1":      # the step's inputs: voxel means (xyz), second frame moved by the motion
    def means(p):
        v, c, n, nv = capi.voxelize(torch.from_numpy(np.ascontiguousarray(p, np.float32)).cuda(), S.PC_RANGE, S.VOXEL_SIZE,
                                    (1408, 768, 40), S.MAX_POINTS_PER_VOXEL, int(os.environ.get('MAX_VOXELS', 80000)))
        k = int(nv.item())
        return capi.vfe_mean(v[:k], n[:k]).cpu().numpy()
    pairs = [(means(a), means(b)) for a, b in pairs]
cnt = [min(len(a), len(b)) for a, b in pairs]
N = max(cnt)
A = np.zeros((4, N, 3), np.float32); Bm = np.zeros((4, N, 3), np.float32)
for i, (a, b) in enumerate(pairs):
    A[i, :cnt[i]] = a[:cnt[i], :3]; Bm[i, :cnt[i]] = b[:cnt[i], :3]
a, b = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda()
counts = torch.tensor(cnt, dtype=torch.int32, device="cuda")
d0, i0 = capi.chamfer_nn(a, b, ncnt=counts, mcnt=counts, method="brute")
d1, i1 = capi.chamfer_nn(a, b, ncnt=counts, mcnt=counts, method="grid")
ok = all(torch.equal(d0[k, :cnt[k]], d1[k, :cnt[k]]) and torch.equal(i0[k, :cnt[k]], i1[k, :cnt[k]]) for k in range(4))
print("grid == exhaustive (bits, indices):", ok, flush=True)
for _ in range(3): capi.chamfer_nn(a, b, ncnt=counts, mcnt=counts, method="grid")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    capi.chamfer_nn(a, b, ncnt=counts, mcnt=counts, method="grid")
e1.record(); torch.cuda.synchronize()
print("pruned search 4 x %d x %d: %.1f us per call (all kernels of rslo_chamfer_grid_nn)" % (N, N, 1e3 * e0.elapsed_time(e1) / 20))
if os.environ.get("SWEEP_SEGMENTS", "0") == "1":
    # misaligned pairs too: the second cloud moved by a bad pose (what a random-init head predicts)
    for tag, bb in (("aligned", b), ("shift 9 m / yaw 40 deg", None)):
        if bb is None:
            th = np.deg2rad(40.0)
            R = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=torch.float32, device="cuda")
            bb = (b @ R.T + torch.tensor([9.0, -1.8, 1.2], device="cuda")).contiguous()
        for seg in (0, 1, 2, 4, 8):
            with capi.tuning(chamfer_segments=seg):
                for _ in range(3): capi.chamfer_nn(a, bb, ncnt=counts, mcnt=counts, method="grid")
                e0.record()
                for _ in range(20): capi.chamfer_nn(a, bb, ncnt=counts, mcnt=counts, method="grid")
                e1.record(); torch.cuda.synchronize()
            print("%-24s segments=%d: %.1f us per call" % (tag, seg, 1e3 * e0.elapsed_time(e1) / 20))
