"""1x1 output convolutions of the head: one thread per pixel (conv1x1_split = 0) against a pixel's channels dealt to four waves
(1): us per launch back to back, B = 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
B, N = 4, 50


def t(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


for cin, cout, H, W in [(64, 7, 96, 176), (32, 1, 96, 176), (64, 7, 48, 88), (64, 7, 24, 44)]:
    x = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda")
    b = torch.randn(cout, device="cuda")
    g = torch.randn(B, cout, H, W, device="cuda")
    row = []
    for split in (0, 1):
        with capi.tuning(conv1x1_split=split):
            row.append("split=%d: fwd %5.1f dgrad %5.1f" % (split, t(lambda: capi.conv1x1_fwd(x, w, b)), t(lambda: capi.conv1x1_dgrad(g, w))))
    print("%d->%d @%dx%d (%.1f MB): " % (cin, cout, H, W, x.numel() * 4e-6) + " | ".join(row))
