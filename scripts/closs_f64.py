"""Is the GPU/CPU gradient gap of the consistency loss rounding on an ill-conditioned problem or a bug?  Takes the
inputs the loss receives inside a real step, evaluates (a) the fused HIP kernels, (b) the torch formulation in fp32 on
the CPU, (c) the same formulation in float64, and prints every gradient's distance to (c)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rslo_amd  # noqa: F401,E402
from oracle import cpu_backend  # noqa: E402
from rslo_amd import workload  # noqa: E402


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


def main():
    from test_gpu_model import reduced_pair, trained_like_init
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    trained_like_init(net)
    ex = workload.make_example(net, [list(reduced_pair(1, rings=int(os.environ.get("RINGS", "16")))[:2])])
    closs = net._consistency_loss
    cap = {}
    orig = closs.pair_losses

    def rec(*a, **k):
        cap["a"], cap["k"] = a, k
        return orig(*a, **k)
    closs.pair_losses = rec
    net(ex)["loss"].backward()
    closs.pair_losses = orig
    a, k = list(cap["a"]), dict(cap["k"])
    order = ["xyz_pred", "xyz_target", "cov_pred", "cov_target", "R_pred", "t_pred", "normal_pred", "normal_target"]
    for n in order[len(a):]:
        a.append(k.pop(n))
    names = ["xyz_pred", "xyz_target", "cov_pred", "cov_target", "R_pred", "t_pred", "normal_pred", "normal_target"]
    print("kwargs:", {kk: (v if not isinstance(v, torch.Tensor) else tuple(v.shape)) for kk, v in k.items()})
    cov = a[2].detach()
    print("cov params: lam increments min %s  max %s" % (cov[..., :3].amin((0, 1)).tolist(), cov[..., :3].amax((0, 1)).tolist()))
    res = {}
    for tag, dev, dt in (("gpu", "cuda", torch.float32), ("cpu32", "cpu", torch.float32), ("cpu64", "cpu", torch.float64)):
        args = [x.detach().to(dev, dt).requires_grad_(i < 4) if isinstance(x, torch.Tensor) else x for i, x in enumerate(a)]
        kw = {kk: (v.to(dev) if isinstance(v, torch.Tensor) else v) for kk, v in k.items()}
        if dev == "cpu":
            kw.pop("counts", None)
            with cpu_backend.patched():
                l, rr, tt = closs.cpu().pair_losses(*args, **kw)
        else:
            l, rr, tt = closs.cuda().pair_losses(*args, **kw)
        l.sum().backward()
        res[tag] = (l.detach(), [x.grad for x in args[:4]])
        print(tag, "loss", l.detach().cpu().numpy())
    for i in range(4):
        g64 = res["cpu64"][1][i]
        if g64 is None:
            continue
        print("%-12s gpu-vs-f64 max %.2e rms %.2e | cpu32-vs-f64 max %.2e rms %.2e | gpu-vs-cpu32 max %.2e rms %.2e" %
              ((names[i],) + err(res["gpu"][1][i], g64) + err(res["cpu32"][1][i], g64) +
               err(res["gpu"][1][i], res["cpu32"][1][i])))


if __name__ == "__main__":
    main()
