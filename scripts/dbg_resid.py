"""rslo_cov_residual_bwd (sorted keys, fixed-order sums) twice -> identical bits, and its time per call.  (The atomic form it was
compared with in round 5 is deleted.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
torch.manual_seed(0)
B, N, M = 4, 35000, 35400
p1 = torch.randn(B, N, 3, device="cuda"); tgt = torch.randn(B, M, 3, device="cuda")
cov1 = torch.randn(B, N, 7, device="cuda") * 0.1; cov2 = torch.randn(B, M, 7, device="cuda") * 0.1
idx = torch.randint(0, M, (B, N), device="cuda", dtype=torch.int32)
idx[:, :2000] = 7; idx[1, 5000:5040] = 11
dist = torch.rand(B, N, device="cuda"); thr = torch.full((B,), 0.9, device="cuda")
Rd = torch.eye(3, device="cuda").reshape(1, 9).repeat(B, 1).contiguous()
gloss = torch.ones(B, device="cuda"); cnt = torch.full((B,), 1000.0, device="cuda")
outs = []
for ordered in (1, 1):
    if True:
        o = capi.cov_residual_bwd(p1, tgt, cov1, cov2, idx, dist, thr, Rd, gloss, cnt, 0.005, need_gp1=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            capi.cov_residual_bwd(p1, tgt, cov1, cov2, idx, dist, thr, Rd, gloss, cnt, 0.005, need_gp1=True)
        e1.record(); torch.cuda.synchronize()
        print("ordered=%d: %.1f us per call" % (ordered, e0.elapsed_time(e1) / 20 * 1e3))
        outs.append(o)
for k, name in enumerate(("gp1", "gtgt", "gcov1", "gcov2")):
    a, b = outs[0][k], outs[1][k]
    print(name, "twice identical:", torch.equal(a, b))
