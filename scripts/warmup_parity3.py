"""Diagnostic: ReLU-mask agreement inside BasicBlock blocks.1.1 / 2.2 (HIP fp32 vs float64), weighted by the gradient."""
import os, sys, copy
os.environ["RSLO_FUSED_BLOCK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import rslo_amd  # noqa: F401
from rslo_amd import workload
from oracle import cpu_backend
import test_gpu_model as T

torch.manual_seed(7)
net, _ = workload.build_network()
net.train()
pool = [list(T.reduced_pair(i)[:2]) for i in range(6)]
net.global_step.fill_(50)
ex = workload.make_example(net, [pool[1]])


def instrument(n, store):
    for si, stage in enumerate(n.odom_predictor.blocks):
        for bi, blk in enumerate(stage):
            for bn_name in ("bn1", "bn2"):
                m = getattr(blk, bn_name)
                orig = m.forward

                def fwd(*a, _orig=orig, _name="blocks.%d.%d.%s" % (si, bi, bn_name), **k):
                    out = _orig(*a, **k)
                    t = out[0] if isinstance(out, (list, tuple)) else out
                    if torch.is_tensor(t) and t.requires_grad:
                        store[_name + ".y"] = t.detach().cpu().double()
                        t.register_hook(lambda g, nn=_name: store.__setitem__(nn + ".dy", g.detach().cpu().double()))
                    return out
                m.forward = fwd


nets = [copy.deepcopy(net), copy.deepcopy(net).cpu().double()]
stores = [{}, {}]
for n, s in zip(nets, stores):
    instrument(n, s)
nets[0](ex)["loss"].backward()
ex_cpu = T.example_to_cpu(ex)
with cpu_backend.patched():
    nets[1](T.example_to_f64(ex_cpu))["loss"].backward()
for key in sorted(k for k in stores[1] if k.endswith(".y")):
    base = key[:-2]
    yg, yd = stores[0].get(key), stores[1][key]
    dg, dd = stores[0].get(base + ".dy"), stores[1].get(base + ".dy")
    if yg is None or dd is None:
        continue
    flip = (yg > 0) != (yd > 0)
    w = dd.abs()
    print("%-18s  y err %.1e | mask flips %5d of %8d | |dy| on flipped pixels / total |dy| = %.2e | max |y| at a flip %.1e | dy err max %.1e l2 %.1e" % (
        base, float((yg - yd).abs().max() / yd.abs().max()), int(flip.sum()), flip.numel(), float((w * flip).sum() / w.sum()),
        float(torch.where(flip, yd.abs(), torch.zeros_like(yd)).max()), float((dg - dd).abs().max() / dd.abs().max()),
        float((dg - dd).norm() / dd.norm())))
