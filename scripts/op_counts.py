"""torch.profiler view of one C3 step: aten-op counts / host self time, and host time per phase (no syncs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from torch.profiler import profile, ProfilerActivity, record_function
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=8e-5, fused=True)
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
host = {}
def phase(name):
    class P:
        def __enter__(s): s.t = time.perf_counter(); s.r = record_function(name); s.r.__enter__()
        def __exit__(s, *a): s.r.__exit__(*a); host[name] = host.get(name, 0) + time.perf_counter() - s.t
    return P()
def wrapf(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        with phase(label): return f(*a, **k)
    setattr(obj, name, w)
wrapf(net.middle_feature_extractor, "forward", "P:encoder_fwd")
wrapf(net.odom_predictor, "forward", "P:head_fwd")
wrapf(net, "create_loss", "P:loss_fwd")
def step():
    with phase("P:voxelize"): ex = workload.make_example(net, clouds)
    opt.zero_grad(set_to_none=True)
    with phase("P:net_fwd_total"): ret = net(ex)
    with phase("P:backward"): ret["loss"].mean().backward()
    with phase("P:clip+adam"):
        torch.nn.utils.clip_grad_norm_(params, 10.0); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); host.clear()
N = 3
t0 = time.perf_counter()
for _ in range(N): step()
th = time.perf_counter() - t0
torch.cuda.synchronize(); tw = time.perf_counter() - t0
print("no profiler: host-issue %.2f ms/step, wall %.2f ms/step" % (1e3 * th / N, 1e3 * tw / N))
for k, v in host.items(): print("  %-18s host %.2f ms/step" % (k, 1e3 * v / N))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)[:60]
print("%-60s %8s %10s %10s" % ("op", "n/step", "selfcpu ms", "cuda ms"))
for e in rows:
    print("%-60s %8.1f %10.3f %10.3f" % (e.key[:60], e.count / N, e.self_cpu_time_total / N / 1e3, getattr(e, "self_device_time_total", 0) / N / 1e3))
print("total ops/step", sum(e.count for e in ka) / N)

print("---- by device time")
rows = sorted(ka, key=lambda e: -getattr(e, "self_device_time_total", 0))[:45]
for e in rows:
    print("%-60s %8.1f %10.3f" % (e.key[:60], e.count / N, getattr(e, "self_device_time_total", 0) / N / 1e3))
