"""Per-layer timing of the BEV head's 3x3 weight gradients: rslo_conv2d_wgrad vs the library path (MIOpen through
torch.ops.aten.convolution_backward), C3 batch (4 pairs).  Usage: python scripts/bench_conv2d_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd  # noqa: F401
from rslo_amd import capi
import _tuning; _tuning.apply_from_env()

LAYERS = [  # (count per step, cin, cout, H, W, stride)
    (5, 128, 128, 48, 88, 1), (1, 192, 64, 96, 176, 1), (3, 64, 64, 96, 176, 1), (9, 128, 128, 24, 44, 1),
    (9, 256, 256, 12, 22, 1), (1, 256, 128, 96, 176, 2), (3, 64, 32, 96, 176, 1), (1, 128, 128, 48, 88, 1),
    (1, 512, 128, 24, 44, 1), (1, 256, 64, 48, 88, 1), (1, 128, 128, 48, 88, 2), (1, 128, 128, 24, 44, 1),
    (1, 256, 256, 12, 22, 1), (1, 128, 256, 24, 44, 2), (1, 128, 64, 24, 44, 1), (1, 64, 32, 48, 88, 1),
    (1, 32, 64, 48, 88, 1), (1, 64, 64, 24, 44, 1)]
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


tot_mine = tot_lib = 0.0
for cnt, cin, cout, H, W, s in LAYERS:
    x = torch.randn(B, cin, H, W, device="cuda")
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    g = torch.randn(B, cout, Ho, Wo, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda")
    lib = lambda: torch.ops.aten.convolution_backward(g, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                      [False, True, False])
    mine = lambda: capi.conv2d_wgrad(x, g, s)
    a = mine(); b = lib()[1]
    rel = float((a - b).abs().max() / b.abs().max())
    tm, tl = timeit(mine), timeit(lib)
    gf = 2.0 * B * Ho * Wo * cin * cout * 9 / 1e9
    print("%dx %3d->%3d %3dx%3d s%d: mine %7.1f us (%6.1f TF)  lib %7.1f us (%6.1f TF)  rel diff %.1e" %
          (cnt, cin, cout, H, W, s, tm, gf / tm * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (tm * 1e-6) / 1e3, tl, gf / (tl * 1e-6) / 1e3, rel), flush=True)
    tot_mine += cnt * tm; tot_lib += cnt * tl
print("per step: mine %.2f ms, library %.2f ms" % (tot_mine / 1e3, tot_lib / 1e3))
