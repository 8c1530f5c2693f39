"""rslo_bn2d_fwd_local / _bwd_local on the head's map sizes (B = 4), N launches each, for rocprofv3 kernel traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
for C, H, W in [(128, 48, 88), (128, 24, 44), (256, 12, 22), (64, 96, 176)]:
    x = torch.randn(4, C, H, W, device="cuda"); g = torch.randn_like(x)
    w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    for _ in range(10):
        y, mean, invstd = capi.bn2d_fwd_local(x, None, w, b, rm, rv, 0.1, 1e-5, 0.0)
        capi.bn2d_bwd_local(g, y, x, w, mean, invstd, 0.0, True, False)
torch.cuda.synchronize()
