"""Error report of the assembled path, GPU (HIP kernels) vs CPU (same host modules over the oracle backend):
poses, loss terms and the per-parameter gradient errors, at a chosen size.  Used to set the bars of
tests/test_gpu_model.py from evidence.

    python scripts/parity_report.py [--bs 4] [--rings 64] [--seed 7] [--init default|trained]
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rslo_amd  # noqa: F401,E402
from oracle import cpu_backend  # noqa: E402
from rslo_amd import synthetic, workload  # noqa: E402


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def relrms(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--init", default="trained")
    ap.add_argument("--step", type=int, default=2000)
    a = ap.parse_args()
    from test_gpu_model import reduced_pair, example_to_cpu, bias_before_bn, trained_like_init
    torch.manual_seed(a.seed)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(a.step)
    if a.init == "trained":
        trained_like_init(net)
    clouds = []
    for b in range(a.bs):
        p0, p1, _ = reduced_pair(b + 1, rings=a.rings)
        clouds.append([p0, p1])
    ex = workload.make_example(net, clouds)
    print("voxels per frame:", [int(v.sum()) for v in ex["num_voxels"]])
    net_cpu = copy.deepcopy(net).cpu()
    t0 = time.time()
    ret = net(ex)
    ret["loss"].backward()
    torch.cuda.synchronize()
    t1 = time.time()
    with cpu_backend.patched():
        ret_c = net_cpu(example_to_cpu(ex))
        ret_c["loss"].backward()
    t2 = time.time()
    from test_gpu_model import example_to_f64
    net64 = copy.deepcopy(net).cpu().double()
    net64.zero_grad()
    with cpu_backend.patched():
        ret64 = net64(example_to_f64(example_to_cpu(ex)))
        ret64["loss"].backward()
    t3 = time.time()
    print("gpu %.2fs cpu %.1fs cpu-f64 %.1fs" % (t1 - t0, t2 - t1, t3 - t2))
    for k in ("translation_preds", "rotation_preds", "loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
        print("%-20s gpu-cpu32 %.3e  gpu-f64 %.3e  cpu32-f64 %.3e   gpu %s" % (k, rel(ret[k], ret_c[k]), rel(ret[k], ret64[k]), rel(ret_c[k], ret64[k]), ret[k].detach().cpu().numpy().reshape(-1)[:8].round(5)))
    skip = bias_before_bn(net)
    rows = []
    for (n, p), (_, pc), (_, p64) in zip(net.named_parameters(), net_cpu.named_parameters(), net64.named_parameters()):
        if n in skip or pc.grad is None or p.grad is None:
            continue
        rows.append((rel(p.grad, pc.grad), rel(p.grad, p64.grad), rel(pc.grad, p64.grad), float(pc.grad.abs().max()), n))
    rows.sort(reverse=True)
    print("params compared:", len(rows))
    for r in rows[:25]:
        print("gpu-cpu32 %.3e  gpu-f64 %.3e  cpu32-f64 %.3e  |g|max %.3e  %s" % r)
    big = [r for r in rows if r[3] > 1e-6]
    ratio = np.array([r[1] / max(r[2], 1e-12) for r in big])
    print("ratio (gpu-f64)/(cpu32-f64): median %.2f p90 %.2f max %.2f (%s)" % (
        np.median(ratio), np.percentile(ratio, 90), ratio.max(), big[int(ratio.argmax())][4]))
    print("gpu-f64: median %.3e max %.3e | cpu32-f64: median %.3e max %.3e" % (
        np.median([r[1] for r in big]), max(r[1] for r in big), np.median([r[2] for r in big]), max(r[2] for r in big)))
    arr = np.array([r[0] for r in rows])
    print("median %.3e  p90 %.3e  max %.3e" % (np.median(arr), np.percentile(arr, 90), arr.max()))
    for pre in ("middle_feature_extractor", "odom_predictor", "_"):
        sub = np.array([r[0] for r in rows if r[4].startswith(pre)])
        if len(sub):
            print("  %-28s n=%3d median %.3e max %.3e" % (pre, len(sub), np.median(sub), sub.max()))


if __name__ == "__main__":
    main()
