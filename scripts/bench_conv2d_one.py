"""One shape of rslo_conv2d_fwd, launched N times (for rocprofv3 --kernel-trace / --pmc runs: kernel durations without
launch overhead).  SHAPE=cin,cout,H,W [B=4] [N=20]; tile configuration through RSLO_TUNING (scripts/_tuning.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd  # noqa: F401
from rslo_amd import capi
import _tuning; _tuning.apply_from_env()
cin, cout, H, W = [int(v) for v in os.environ.get("SHAPE", "128,128,48,88").split(",")]
B, N = int(os.environ.get("B", "4")), int(os.environ.get("N", "20"))
x = torch.randn(B, cin, H, W, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
ws = capi.conv2d_wsplit(w, False)
ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
PLANES = os.environ.get("PLANES", "0") == "1"       # k_conv2d_fwd_p on pre-split operand planes
pl = capi.opl_from_nchw(x) if PLANES else None
run = (lambda: capi.conv2d_fwd_p(pl, ws, None, cout)) if PLANES else (lambda: capi.conv2d_fwd(x, ws, None, cout))
y = run()
print("rel err", float((y - ref).abs().max() / ref.abs().max()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    run()
e1.record(); torch.cuda.synchronize()
gf = 2.0 * B * H * W * cin * cout * 9 / 1e9
t = e0.elapsed_time(e1) / N * 1e3
print("%d->%d %dx%d B=%d: %.1f us per launch back-to-back (%.1f TF)" % (cin, cout, H, W, B, t, gf / t * 1e3 / 1e3))
