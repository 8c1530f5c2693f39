"""k_conv2d_wl (weights through LDS, csrc/conv2d_wl.hip) against k_conv2d_fwd on the stride-1 3x3 layers of the BEV head,
C3 batch (4 pairs): us per launch back-to-back, forward and data gradient, weights split beforehand.
Usage: python scripts/bench_conv2d_wl.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd  # noqa: F401
from rslo_amd import capi

LAYERS = [  # (count per step fwd+dgrad, cin, cout, H, W)
    (12, 128, 128, 48, 88), (1, 192, 64, 96, 176), (1, 64, 192, 96, 176), (6, 64, 64, 96, 176), (3, 64, 32, 96, 176), (3, 32, 64, 96, 176),
    (1, 256, 64, 48, 88), (1, 64, 256, 48, 88), (22, 128, 128, 24, 44), (20, 256, 256, 12, 22), (1, 512, 128, 24, 44), (1, 128, 512, 24, 44)]
B = 4


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


modes = [int(m) for m in os.environ.get("MODES", "0,1,3").split(",")]
tot = {m: 0.0 for m in modes}
for cnt, cin, cout, H, W in LAYERS:
    x = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    ws = capi.conv2d_wsplit(w, False)
    out, ref = {}, None
    for m in modes:
        with capi.tuning(conv2d_fwd_wl=m, **({'conv2d_fwd_kc': 1} if (m == 0 and os.environ.get('KC1')) else {})):
            y = capi.conv2d_fwd(x, ws, None, cout).clone()
            ref = y if ref is None else ref
            same = bool(torch.equal(y, ref))
            out[m] = (timeit(lambda: capi.conv2d_fwd(x, ws, None, cout)), same)
            tot[m] += cnt * out[m][0]
    gf = 2.0 * B * H * W * cin * cout * 9 / 1e9
    print("%2dx %3d->%3d %3dx%3d: " % (cnt, cin, cout, H, W) +
          " | ".join("wl=%d %6.1f us (%5.1f TF)%s" % (m, out[m][0], gf / out[m][0] * 1e3 / 1e3, "" if out[m][1] else " BITS DIFFER")
                     for m in modes), flush=True)
print("per step: " + ", ".join("wl=%d %.3f ms" % (m, tot[m] / 1e3) for m in modes))
