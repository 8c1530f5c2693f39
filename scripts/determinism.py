"""Run the same training forward several times (fresh identical nets) and report loss spread; optional prefetch path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rslo_amd
from rslo_amd import workload, synthetic
def reduced_pair(seed=0, rings=16):
    p0, p1, motion = synthetic.frame_pair(seed)
    r0 = np.arange(len(p0)) * 64 // len(p0); r1 = np.arange(len(p1)) * 64 // len(p1)
    step = 64 // rings
    return p0[r0 % step == 0], p1[r1 % step == 0]
pairs = [reduced_pair(7), reduced_pair(8, rings=32), reduced_pair(9)]
batches = [[[torch.from_numpy(p[0]).cuda(), torch.from_numpy(p[1]).cuda()]] for p in pairs]
mode = sys.argv[1] if len(sys.argv) > 1 else "inline"
for rep in range(4):
    torch.manual_seed(9)
    net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
    out = []
    if mode == "prefetch":
        pf = workload.ExamplePrefetcher(net); pf.submit(batches[0])
    for i, clouds in enumerate(batches):
        ex = pf.get() if mode == "prefetch" else workload.make_example(net, clouds)
        r = net(ex)
        if mode == "prefetch" and i + 1 < len(batches): pf.submit(batches[i + 1])
        r["loss"].mean().backward()
        out.append("%.6f/%.6f/%.6f" % (float(r["loss"]), float(r["C_loss"]), float(r["pyramid_loss"])))
        net.zero_grad(set_to_none=True)
    if mode == "prefetch": pf.close()
    print(mode, rep, " ".join(out), flush=True)
