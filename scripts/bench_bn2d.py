"""rslo_bn2d_fwd_local / _bwd_local on the head's map sizes (B = 4): microseconds per launch back-to-back, vectorised
register-cached kernels (bn_small_rc = 1) vs the scalar ones of round 4 (2) vs the two-pass loops (0); results compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, H, W in [(128, 48, 88), (128, 24, 44), (256, 12, 22), (64, 96, 176)]:
    x = torch.randn(4, C, H, W, device="cuda"); g = torch.randn_like(x); res = torch.randn_like(x)
    w = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda")
    outs, line = {}, "%3d ch %3dx%3d:" % (C, H, W)
    for mode in (1, 2, 0):
        with capi.tuning(bn_small_rc=mode):
            rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
            y, mean, invstd = capi.bn2d_fwd_local(x, res, w, b, rm, rv, 0.1, 1e-3, 0.0)
            dx, dres, dg, db = capi.bn2d_bwd_local(g, y, x, w, mean, invstd, 0.0, True, True)
            outs[mode] = (y, mean, invstd, dx, dres, dg, db)
            tf = timeit(lambda: capi.bn2d_fwd_local(x, res, w, b, rm, rv, 0.1, 1e-3, 0.0))
            tb = timeit(lambda: capi.bn2d_bwd_local(g, y, x, w, mean, invstd, 0.0, True, True))
            line += "  rc=%d fwd %5.1f bwd %5.1f us" % (mode, tf, tb)
    err = max(float((a - c).abs().max() / (c.abs().max() + 1e-12)) for a, c in zip(outs[1], outs[2]))
    print(line + "  | vectorised vs scalar max rel diff %.1e" % err, flush=True)
