"""Host cost of replaying the head's graphs (rslo_amd/headgraph.py): wall time of every piece of a step's graph path on the
issuing threads, with the GPU drained in front of the replay or not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import workload, synthetic, headgraph
torch.manual_seed(7)
net, _ = workload.build_network()
net.train()
B = int(os.environ.get("B", "4"))
clouds = [list(synthetic.frame_pair(b)[:2]) for b in range(B)]
ex = workload.make_example(net, clouds)
if os.environ.get("STREAM") == "1":
    torch.cuda.set_stream(torch.cuda.Stream(priority=min(torch.cuda.Stream.priority_range())))
T = {}
orig_replay = headgraph.HeadGraph.replay
EV = {}
def timed(g, name):
    if os.environ.get("DRAIN") == "1":
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter(); g.replay(); T.setdefault(name, []).append(time.perf_counter() - t0)
    e1.record()
    EV.setdefault(name, []).append((e0, e1))
headgraph.HeadGraph.replay = staticmethod(timed)
GE = []
for i in range(12):
    net.zero_grad(set_to_none=True)
    g0, g1, g2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    g0.record()
    ret = net(ex)
    g1.record()
    t1 = time.perf_counter()
    ret["loss"].mean().backward()
    g2.record()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("   gpu: forward+loss %.2f ms, backward %.2f ms" % (g0.elapsed_time(g1), g1.elapsed_time(g2)))
    print("step %d: fwd host %.2f ms, bwd host %.2f ms, drain %.2f ms" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
for k, v in T.items():
    print(k, "replay host ms:", ["%.2f" % (x * 1e3) for x in v])
for k, v in EV.items():
    print(k, "replay gpu ms:", ["%.2f" % a.elapsed_time(b) for a, b in v])
if not headgraph.ENABLED:
    sys.exit(0)
hg = headgraph._STATE[net.odom_predictor].graph
print("outputs", len(hg.flat), "requiring grad", len(hg.req), "param grads", len(hg.pgrads))
