"""Diagnostic: three-way gradient comparison (HIP fp32 / CPU-oracle fp32 / CPU-oracle float64) in the WARM-UP regime
(global_step <= 1500: identity pose inside the consistency loss, icp_iter = 5) on default-init weights, optionally after
N real optimizer steps.  Prints per-module medians and the worst tensors.  usage: warmup_parity.py [n_train_steps] [global_step]"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import rslo_amd  # noqa: F401
from rslo_amd import workload
import test_gpu_model as T

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gstep = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.manual_seed(7)
net, _ = workload.build_network()
net.train()
pool = [list(T.reduced_pair(i)[:2]) for i in range(6)]
if n_train:
    T.real_training_steps(net, n_train, lambda i: [pool[(2 * i) % 6], pool[(2 * i + 1) % 6]])
net.global_step.fill_(gstep)
net.zero_grad(set_to_none=True)
ex = workload.make_example(net, [pool[1]])
(ret, g), (ret_c, c), (ret_64, d) = T.three_way(copy.deepcopy(net), ex)
for k in ("translation_preds", "rotation_preds", "loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
    print("%-18s gpu-cpu %.2e  gpu-f64 %.2e  cpu-f64 %.2e   %s" % (k, T.rel(ret[k], ret_c[k]), T.rel(ret[k], ret_64[k]),
                                                                  T.rel(ret_c[k], ret_64[k]), ret_64[k].detach().flatten()[:4].tolist()))
rows = T.gradient_errors([g, c, d], T.bias_before_bn(net))
by = {}
for e_g, e_c, n, l2 in rows:
    key = ".".join(n.split(".")[:3])
    by.setdefault(key, []).append((e_g, e_c))
for k, v in sorted(by.items()):
    a = np.array(v)
    print("%-60s n=%3d  gpu median %.2e max %.2e | cpu median %.2e max %.2e" % (k, len(v), np.median(a[:, 0]), a[:, 0].max(),
                                                                              np.median(a[:, 1]), a[:, 1].max()))
print("worst 15:")
for e_g, e_c, n, l2 in sorted(rows, reverse=True)[:15]:
    print("  %.2e  %.2e  l2 %.2e  %s" % (e_g, e_c, l2, n))
