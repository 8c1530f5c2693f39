"""Which modules / loss pieces launch the most GPU ops in the FORWARD of one C3 step (profiler event tree;
backward ops mirror the forward ones).  Labels: M:<module path> via forward hooks, F:<function> via wrappers."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from torch.profiler import profile, ProfilerActivity, record_function
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=8e-5, fused=True)
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
DEPTH = int(os.environ.get("DEPTH", "3"))
for name, m in net.named_modules():
    if name and name.count(".") < DEPTH:
        st = []
        m.register_forward_pre_hook(lambda mod, a, n=name, st=st: st.append(record_function("M:" + n).__enter__()) or None)
        m.register_forward_hook(lambda mod, a, o, st=st: st.pop().__exit__(None, None, None) or None)
def wrapf(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        with record_function("F:" + label): return f(*a, **k)
    setattr(obj, name, w)
wrapf(net, "create_loss", "create_loss"); wrapf(net, "gen_tq_maps", "gen_tq_maps")
wrapf(net._consistency_loss, "pair_losses", "pair_losses")
wrapf(net._translation_loss, "_compute_loss", "AdaptiveL2.T"); wrapf(net._rotation_loss, "_compute_loss", "AdaptiveL2.R")
wrapf(net.odom_predictor, "vote", "vote"); wrapf(net, "network_forward", "network_forward")
wrapf(workload, "make_example", "make_example")
def step():
    ex = workload.make_example(net, clouds)
    opt.zero_grad(set_to_none=True)
    ret = net(ex)
    with record_function("F:backward"): ret["loss"].mean().backward()
    with record_function("F:clip+adam"):
        torch.nn.utils.clip_grad_norm_(params, 10.0); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
cnt, dev = collections.Counter(), collections.Counter()
ops = collections.defaultdict(collections.Counter)       # label -> top-level aten op under it -> launches
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks: continue
    p, lab, top = e, None, e.name
    while p is not None:
        if p.name.startswith(("M:", "F:")): lab = p.name; break
        top = p.name
        p = p.cpu_parent
    lab = lab or "<none>"
    cnt[lab] += len(ks); dev[lab] += sum(k.duration for k in ks)
    ops[lab][top] += len(ks)
print("%-70s %6s %9s" % ("innermost label", "n", "gpu ms"))
for s, v in sorted(dev.items(), key=lambda kv: -kv[1])[:90]:
    print("%-70s %6d %9.3f" % (s[:70], cnt[s], v / 1e3))
print("total gpu-op count", sum(cnt.values()), "gpu ms", sum(dev.values()) / 1e3)

for lab in os.environ.get("OPS_OF", "").split(","):
    if lab in ops:
        print("== launches under", lab)
        for k, v in ops[lab].most_common(40):
            print("  %4d  %s" % (v, k[:100]))
