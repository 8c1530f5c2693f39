"""Per-layer timing of the BEV head's 3x3 stride-1 forward / data-gradient passes: rslo_conv2d_fwd (+ the weight split)
vs the library path (MIOpen through torch), C3 batch (4 pairs).  Usage: python scripts/bench_conv2d_fwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd  # noqa: F401
from rslo_amd import capi

LAYERS = [  # (count per step, cin, cout, H, W)
    (5, 128, 128, 48, 88), (1, 192, 64, 96, 176), (3, 64, 64, 96, 176), (9, 128, 128, 24, 44),
    (9, 256, 256, 12, 22), (3, 64, 32, 96, 176), (1, 128, 128, 48, 88),
    (1, 512, 128, 24, 44), (1, 256, 64, 48, 88), (1, 128, 128, 24, 44),
    (1, 256, 256, 12, 22), (1, 128, 64, 24, 44), (1, 64, 32, 48, 88),
    (1, 32, 64, 48, 88), (1, 64, 64, 24, 44)]
B = 4


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0] * 4
for cnt, cin, cout, H, W in LAYERS:
    x = torch.randn(B, cin, H, W, device="cuda")
    g = torch.randn(B, cout, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    ws, wst = capi.conv2d_wsplit(w, False), capi.conv2d_wsplit(w, True)
    f_lib = lambda: torch.nn.functional.conv2d(x, w, None, 1, 1)
    d_lib = lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [True, False, False])
    presplit = os.environ.get('PRESPLIT', '0') == '1'     # time the conv kernel alone (weights split beforehand)
    f_mine = (lambda: capi.conv2d_fwd(x, ws, None, cout)) if presplit else (lambda: capi.conv2d_fwd(x, capi.conv2d_wsplit(w, False), None, cout))
    d_mine = (lambda: capi.conv2d_fwd(g, wst, None, cin)) if presplit else (lambda: capi.conv2d_fwd(g, capi.conv2d_wsplit(w, True), None, cin))
    rf = float((f_mine() - f_lib()).abs().max() / f_lib().abs().max())
    rd = float((d_mine() - d_lib()[0]).abs().max() / d_lib()[0].abs().max())
    t = [timeit(f_mine), timeit(f_lib), timeit(d_mine), timeit(d_lib)]
    gf = 2.0 * B * H * W * cin * cout * 9 / 1e9
    print("%dx %3d->%3d %3dx%3d: fwd mine %6.1f us (%5.1f TF) lib %6.1f us | dgrad mine %6.1f us (%5.1f TF) lib %6.1f us | rel %.0e %.0e"
          % (cnt, cin, cout, H, W, t[0], gf / t[0] * 1e3 / 1e3, t[1], t[2], gf / t[2] * 1e3 / 1e3, t[3], rf, rd), flush=True)
    for i in range(4):
        tot[i] += cnt * t[i]
print("per step: fwd mine %.2f ms, lib %.2f ms; dgrad mine %.2f ms, lib %.2f ms" % tuple(v / 1e3 for v in tot))
