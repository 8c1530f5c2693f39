"""Diagnostic: probe-gradient errors GPU vs float64 arbiter, fp32 mode and bf16 (amp O1) mode, per parameter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import copy
import numpy as np, torch
import rslo_amd
from oracle import cpu_backend
from rslo_amd import workload
import test_gpu_model as T

def run(lp):
    from apex import amp
    torch.manual_seed(7)
    net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000); T.trained_like_init(net)
    ex = workload.make_example(net, [list(T.reduced_pair(1)[:2])])
    net64 = copy.deepcopy(net).cpu().double()
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    try:
        if lp:
            net, opt = amp.initialize(net, opt, opt_level="O1")
        ret, probe = T._probe_step(net, ex)
        cpu_backend.result_dtype = torch.float64
        with cpu_backend.patched():
            ret64, probe64 = T._probe_step(net64, T.example_to_f64(T.example_to_cpu(ex)))
    finally:
        cpu_backend.result_dtype = torch.float32
        amp.initialize(net, opt, opt_level="O0")
    for i, (a, b) in enumerate(zip(ret["_probe_outputs"], ret64["_probe_outputs"])):
        print("  output %d shape %s rel %.3e  max|ref| %.3e" % (i, tuple(a.shape), T.rel(a, b), float(b.abs().max())))
    rows = []
    for (n, p), (_, q) in zip(net.named_parameters(), net64.named_parameters()):
        if q.grad is None or p.grad is None: continue
        rows.append((T.rel(p.grad, q.grad), float(q.grad.abs().max()), n))
    print("lp", lp, "probe", float(probe), float(probe64), "median", np.median([r[0] for r in rows]))
    for r in rows:
        print("  %.3e  max|g| %.3e  %s" % r)
run(True)
