"""Diagnostic: ONE BasicBlock (blocks.1.1 by default) stand-alone on the inputs / output gradient it sees inside the network:
HIP fp32 vs the same block in float64 on the CPU, stage by stage (bn2 bwd -> conv2 dgrad -> bn1 bwd -> conv1 dgrad)."""
import os, sys, copy
os.environ["RSLO_FUSED_BLOCK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import torch.nn.functional as F
import rslo_amd  # noqa: F401
from rslo_amd import workload
import test_gpu_model as T

si, bi = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1.1").split("."))
torch.manual_seed(7)
net, _ = workload.build_network()
net.train()
pool = [list(T.reduced_pair(i)[:2]) for i in range(6)]
net.global_step.fill_(50)
ex = workload.make_example(net, [pool[1]])
blk = net.odom_predictor.blocks[si][bi]
cap = {}
orig = blk.forward


def fwd(x):
    cap["x"] = (x[0] if isinstance(x, (list, tuple)) else x).detach().clone()
    out = orig(x)
    t = out[0] if isinstance(out, (list, tuple)) else out
    t.register_hook(lambda g: cap.__setitem__("gy", g.detach().clone()))
    return out


blk.forward = fwd
net(ex)["loss"].backward()
blk.forward = orig
x, gy = cap["x"], cap["gy"]
print("block %d.%d: x %s |max| %.3e, gy |max| %.3e, gy nonzero frac %.3f" % (si, bi, tuple(x.shape), float(x.abs().max()),
                                                                          float(gy.abs().max()), float((gy != 0).float().mean())))
w1 = getattr(blk.conv1, "conv1", blk.conv1).weight.detach()
w2 = getattr(blk.conv2, "conv1", blk.conv2).weight.detach()
eps = blk.bn1.eps


def run(x, gy, w1, w2, dt):
    """plain torch formulation of the block in dtype dt on the CPU; returns the stage tensors"""
    x = x.cpu().to(dt).requires_grad_(True)
    w1 = w1.cpu().to(dt).requires_grad_(True)
    w2 = w2.cpu().to(dt).requires_grad_(True)
    g1 = blk.bn1.weight.detach().cpu().to(dt).requires_grad_(True)
    b1 = blk.bn1.bias.detach().cpu().to(dt).requires_grad_(True)
    g2 = blk.bn2.weight.detach().cpu().to(dt).requires_grad_(True)
    b2 = blk.bn2.bias.detach().cpu().to(dt).requires_grad_(True)
    o1 = F.conv2d(x, w1, None, 1, 1); o1.retain_grad()
    y1 = F.relu(F.batch_norm(o1, None, None, g1, b1, True, 0.0, eps)); y1.retain_grad()
    o2 = F.conv2d(y1, w2, None, 1, 1); o2.retain_grad()
    y2 = F.relu(F.batch_norm(o2, None, None, g2, b2, True, 0.0, eps) + x)
    y2.backward(gy.cpu().to(dt))
    return {"o1": o1, "y1": y1, "o2": o2, "y2": y2, "d_o2": o2.grad, "d_y1": y1.grad, "d_o1": o1.grad, "dx": x.grad,
            "dw1": w1.grad, "dw2": w2.grad, "db1": b1.grad, "dg1": g1.grad, "db2": b2.grad, "dg2": g2.grad}


ref = run(x, gy, w1, w2, torch.float64)
c32 = run(x, gy, w1, w2, torch.float32)
# the HIP block stand-alone on the same inputs
b2 = copy.deepcopy(blk)
b2.zero_grad(set_to_none=True)
xg = x.clone().requires_grad_(True)
out = b2([xg, None])
yg = out[0] if isinstance(out, (list, tuple)) else out
yg.backward(gy)
hip = {"y2": yg, "dx": xg.grad, "dw1": getattr(b2.conv1, "conv1", b2.conv1).weight.grad,
       "dw2": getattr(b2.conv2, "conv1", b2.conv2).weight.grad, "db1": b2.bn1.bias.grad, "dg1": b2.bn1.weight.grad,
       "db2": b2.bn2.bias.grad, "dg2": b2.bn2.weight.grad}


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


for k in ("y2", "dx", "dw1", "dw2", "db1", "dg1", "db2", "dg2"):
    print("%-4s HIP vs f64: max %.2e l2 %.2e   | torch-fp32 vs f64: max %.2e l2 %.2e   |max ref| %.3e" % (
        (k,) + rel(hip[k], ref[k]) + rel(c32[k], ref[k]) + (float(ref[k].abs().max()),)))
for k in ("d_o2", "d_y1", "d_o1"):
    print("%-4s                                  torch-fp32 vs f64: max %.2e l2 %.2e   |max ref| %.3e" % (
        (k,) + rel(c32[k], ref[k]) + (float(ref[k].abs().max()),)))
torch.save({"x": x.cpu(), "gy": gy.cpu(), "w1": w1.cpu(), "w2": w2.cpu(), "g1": blk.bn1.weight.detach().cpu(),
            "b1": blk.bn1.bias.detach().cpu(), "g2": blk.bn2.weight.detach().cpu(), "b2": blk.bn2.bias.detach().cpu()},
           os.environ.get("DUMP", "/tmp/block_case.pt"))
