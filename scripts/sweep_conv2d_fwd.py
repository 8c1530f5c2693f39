"""Kernel time of rslo_conv2d_fwd per tile configuration (rslo_tuning_set switches) on the head's layer shapes, B = 4:
HIP events around N back-to-back launches, in one process.  usage: python scripts/sweep_conv2d_fwd.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd  # noqa: F401
from rslo_amd import capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SHAPES = [(128, 128, 48, 88), (64, 64, 96, 176), (64, 192, 96, 176), (192, 64, 96, 176), (256, 64, 48, 88),
          (128, 128, 24, 44), (256, 256, 12, 22), (256, 128, 48, 88), (512, 128, 24, 44)]
CFGS = [("default", {}), ("lean0", dict(conv2d_fwd_lean=0)), ("4,2", dict(conv2d_fwd_tr=4, conv2d_fwd_mtw=2)),
        ("4,2,occ4", dict(conv2d_fwd_tr=4, conv2d_fwd_mtw=2, conv2d_fwd_occ=4)),
        ("4,2,occ3", dict(conv2d_fwd_tr=4, conv2d_fwd_mtw=2, conv2d_fwd_occ=3)),
        ("4,1,occ4", dict(conv2d_fwd_tr=4, conv2d_fwd_mtw=1, conv2d_fwd_occ=4)),
        ("8,1", dict(conv2d_fwd_tr=8, conv2d_fwd_mtw=1)), ("8,2", dict(conv2d_fwd_tr=8, conv2d_fwd_mtw=2)),
        ("6,1,occ3", dict(conv2d_fwd_tr=6)), ("6,1,occ4", dict(conv2d_fwd_tr=6, conv2d_fwd_occ=4)), ("6,1,fulla", dict(conv2d_fwd_tr=6, conv2d_fwd_occ=2))]
if os.environ.get("TR6") == "1":
    CFGS = [c for c in CFGS if c[0] in ("default", "8,1") or c[0].startswith("6,")]
CFGS += [(k, v) for k, v in (("cw4", dict(conv2d_fwd_cw=4)), ("cw4,tr2", dict(conv2d_fwd_cw=4, conv2d_fwd_tr=2)))
         if os.environ.get("CW", "0") == "1"]
B = 4
for cin, cout, H, W in SHAPES:
    x = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    ws = capi.conv2d_wsplit(w, False)
    ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
    row = []
    for name, kw in CFGS:
        try:
            with capi.tuning(**kw):
                y = capi.conv2d_fwd(x, ws, None, cout)
                err = float((y - ref).abs().max() / ref.abs().max())
                for _ in range(5):
                    capi.conv2d_fwd(x, ws, None, cout)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(N):
                    capi.conv2d_fwd(x, ws, None, cout)
                e1.record()
                torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / N * 1e3
            row.append("%s %.1f%s" % (name, t, "" if err < 3e-5 else " ERR %.1e" % err))
        except Exception as e:
            row.append("%s FAIL(%s)" % (name, str(e)[:40]))
    gf = 2.0 * B * H * W * cin * cout * 9 / 1e9
    print("%3d->%3d %3dx%3d (%.1f GF): " % (cin, cout, H, W, gf) + " | ".join(row), flush=True)
