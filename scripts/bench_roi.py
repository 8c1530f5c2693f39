import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
g = torch.Generator().manual_seed(1)
d = (torch.rand(4, 34938, generator=g) ** 3 * 40).cuda(); cnt = torch.tensor([31000, 34938, 30011, 33000], dtype=torch.int32).cuda()
for _ in range(3): capi.roi_threshold(d, cnt, 0.97)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): capi.roi_threshold(d, cnt, 0.97)
e1.record(); torch.cuda.synchronize()
print("roi_threshold 4 x 34938: %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
