"""Stride-2 layers of the BEV stages: forward / data gradient of rslo_conv2d_fwd_s2 / _dgrad_s2, N launches each (for
rocprofv3 kernel traces).  B = 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
B, N = 4, 10
for cin, cout, H, W in [(256, 128, 96, 176), (128, 128, 48, 88), (128, 256, 24, 44)]:
    for k in (3, 1):
        x = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") / (k * cin ** 0.5)
        wf, wt = capi.conv2d_wsplit_k(w, False), capi.conv2d_wsplit_k(w, True)
        y = capi.conv2d_fwd_s2(x, wf, cout, k)
        g = torch.randn_like(y)
        ref = torch.nn.functional.conv2d(x, w, None, 2, (k - 1) // 2)
        print(cin, cout, H, W, k, "rel err", float((y - ref).abs().max() / ref.abs().max()))
        for _ in range(N):
            capi.conv2d_fwd_s2(x, wf, cout, k)
            capi.conv2d_dgrad_s2(g, wt, cin, H, W, k)
torch.cuda.synchronize()
