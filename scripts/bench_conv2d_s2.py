"""Stride-2 layers of the BEV stages: forward / data gradient of rslo_conv2d_fwd_s2 / _dgrad_s2 (k_conv2d_str2), one and two
16-channel blocks per wave: us per launch back to back (HIP events), B = 4.  (Round 5 ran it against the run-time-loop kernel
k_conv2d_str as well -- profiles/r05_*; that kernel is deleted.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
B, N = 4, 30


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


for cin, cout, H, W in [(256, 128, 96, 176), (128, 128, 48, 88), (128, 256, 24, 44)]:
    for k in (3, 1):
        x = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") / (k * cin ** 0.5)
        wf, wt = capi.conv2d_wsplit_k(w, False), capi.conv2d_wsplit_k(w, True)
        g = torch.randn(B, cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device="cuda")
        row = []
        for mtw in (0, 1, 2):
            with capi.tuning(conv2d_s2_mtw=mtw):
                row.append("mtw=%d: fwd %6.1f dgrad %6.1f" % (
                    mtw, t(lambda: capi.conv2d_fwd_s2(x, wf, cout, k)), t(lambda: capi.conv2d_dgrad_s2(g, wt, cin, H, W, k))))
        gf = 2.0 * B * g.shape[2] * g.shape[3] * cin * cout * k * k * 1e-9
        print("%d->%d @%dx%d k=%d (%.2f GFLOP, six-product floor %.1f us): " % (cin, cout, H, W, k, gf, gf * 6 / 2.5e3) + " | ".join(row))
