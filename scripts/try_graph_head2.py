"""Staged HIP-graph capture experiment on the dense head: where does capture break?"""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train()
head = net.odom_predictor
B = 4
xs = [(torch.randn(B, 128, 96, 176, device="cuda") * (torch.rand(B, 1, 96, 176, device="cuda") > 0.8)) for _ in range(2)]
stage = sys.argv[1]

def fwd():
    r = head([xs[0], xs[1]])
    out = [r["translation_preds"][0], r["rotation_preds"][0], r["tq_map_g"], r["t_conf"], r["r_conf"]]
    for p, m in r["pyramid_motion"]: out += [p, m]
    return out

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        if stage == "fwd":
            with torch.no_grad(): fwd()
        else:
            for x in xs: x.requires_grad_(True)
            outs = fwd(); loss = sum((o ** 2).mean() for o in outs if o.requires_grad); loss.backward()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("warmup done", flush=True)
if stage != "fwd":      # the same work issued eagerly (what the step does today)
    def eager():
        head.zero_grad(set_to_none=True)
        for x in xs: x.grad = None
        outs = fwd(); loss = sum((o ** 2).mean() for o in outs if o.requires_grad); loss.backward()
    for _ in range(3): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eager()
    torch.cuda.synchronize(); print("eager %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
g = torch.cuda.CUDAGraph()
if stage == "fwd":
    with torch.no_grad():
        with torch.cuda.graph(g):
            outs = fwd()
    print("captured fwd", flush=True)
else:
    head.zero_grad(set_to_none=True)
    for x in xs: x.grad = None
    with torch.cuda.graph(g):
        outs = fwd(); loss = sum((o ** 2).mean() for o in outs if o.requires_grad); loss.backward()
    print("captured fwd+bwd", flush=True)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print("replay %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
