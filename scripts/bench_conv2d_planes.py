"""k_conv2d_fwd (NCHW fp32 staged + split per workgroup) vs k_conv2d_fwd_p (pre-split operand planes), per layer shape of
the BEV head, C3 batch: bit equality and microseconds per launch, plus the stand-alone producer k_opl_from_nchw.
Usage: python scripts/bench_conv2d_planes.py [tr=4|8] [mtw=1|2] [lean=0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd  # noqa: F401
from rslo_amd import capi

LAYERS = [  # (count per step fwd+dgrad, cin, cout, H, W)
    (10, 128, 128, 48, 88), (1, 192, 64, 96, 176), (1, 64, 192, 96, 176), (6, 64, 64, 96, 176), (18, 128, 128, 24, 44),
    (18, 256, 256, 12, 22), (3, 64, 32, 96, 176), (3, 32, 64, 96, 176),
    (1, 512, 128, 24, 44), (1, 128, 512, 24, 44), (1, 256, 64, 48, 88), (1, 64, 256, 48, 88)]
B = 4
cfg = dict(a.split("=") for a in sys.argv[1:])


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tun = {}
if "tr" in cfg: tun["conv2d_fwd_tr"] = int(cfg["tr"])
if "mtw" in cfg: tun["conv2d_fwd_mtw"] = int(cfg["mtw"])
if "lean" in cfg: tun["conv2d_fwd_lean"] = int(cfg["lean"])
tot = [0.0, 0.0, 0.0]
for cnt, cin, cout, H, W in LAYERS:
    x = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    ws = capi.conv2d_wsplit(w, False)
    pl = capi.opl_from_nchw(x)
    ref = capi.conv2d_fwd(x, ws, None, cout)
    with capi.tuning(**tun):
        new = capi.conv2d_fwd_p(pl, ws, None, cout)
        same = bool((ref == new).all())
        err = float((ref - new).abs().max())
        t_new = timeit(lambda: capi.conv2d_fwd_p(pl, ws, None, cout))
    t_old = timeit(lambda: capi.conv2d_fwd(x, ws, None, cout))
    t_spl = timeit(lambda: capi.opl_from_nchw(x))
    gf = 2.0 * B * H * W * cin * cout * 9 / 1e9
    print("%2dx %3d->%3d %3dx%3d: old %6.1f us (%5.1f TF)  planes %6.1f us (%5.1f TF)  split %5.1f us  bits %s (max diff %.1e)"
          % (cnt, cin, cout, H, W, t_old, gf / t_old * 1e3 / 1e3, t_new, gf / t_new * 1e3 / 1e3, t_spl,
             "same" if same else "DIFFER", err), flush=True)
    tot[0] += cnt * t_old; tot[1] += cnt * t_new; tot[2] += cnt * t_spl
print("per step: old %.2f ms, planes %.2f ms, stand-alone split %.2f ms" % tuple(v / 1e3 for v in tot))
