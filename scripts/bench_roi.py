import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
g = torch.Generator().manual_seed(1)
# argv[1]: "spread" (default, rand^3 * 40: leading bytes over many exponents), "narrow" (0.5 .. 1: ONE leading byte, every
# first-pass histogram increment of a wave hits the same LDS bin), "chamfer" (squared NN distances of two jittered clouds)
kind = sys.argv[1] if len(sys.argv) > 1 else "spread"
r = torch.rand(4, 34938, generator=g)
if kind == "narrow":
    d = (r * 0.5 + 0.5).cuda()
elif kind == "chamfer":
    a = torch.rand(4, 34938, 3, generator=g) * torch.tensor([140.0, 80.0, 4.0]); b = a + 0.05 * torch.randn(a.shape, generator=g)
    d = torch.stack([torch.cdist(a[i, :, :].cuda(), b[i, ::2, :].cuda()).min(1).values ** 2 for i in range(4)])
else:
    d = (r ** 3 * 40).cuda()
cnt = torch.tensor([31000, 34938, 30011, 33000], dtype=torch.int32).cuda()
for _ in range(3): capi.roi_threshold(d, cnt, 0.97)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): capi.roi_threshold(d, cnt, 0.97)
e1.record(); torch.cuda.synchronize()
print("roi_threshold 4 x 34938 (%s): %.1f us" % (kind, e0.elapsed_time(e1) / 50 * 1e3))
