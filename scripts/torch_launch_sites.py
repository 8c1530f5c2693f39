"""Which Python lines still launch torch-native GPU kernels (at::native / rocclr copies / library GEMMs) in one C3 step --
the step bench.py times: examples from the ExamplePrefetcher (native planner on the side stream), head graph as configured.
Forward launches are attributed to the innermost frame under rslo_amd/ (with_stack); backward launches to the autograd
node that issued them.  RSLO_SITES_INLINE=1: voxelize + plan inline through workload.make_example instead (the
Python planner the native one is tested against)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from torch.profiler import profile, ProfilerActivity
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
params = [p for p in net.parameters() if p.requires_grad]
from rslo.builder import lr_scheduler_builder, optimizer_builder
from rslo_amd import optim as hip_optim
from rslo.utils import config_text
train_cfg = config_text.shipped_config().train_config
opt = optimizer_builder.build(train_cfg.optimizer, net)
sched = lr_scheduler_builder.build(train_cfg.optimizer, opt, train_cfg.steps)
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
inline = os.environ.get("RSLO_SITES_INLINE") == "1"
prefetch = None
if not inline:
    prefetch = workload.ExamplePrefetcher(net, device="cuda", depth=2, workers=1)
    for _ in range(2): prefetch.submit(clouds)
def step():
    ex = workload.make_example(net, clouds) if inline else prefetch.get()
    sched.step(net.get_global_step())
    opt.zero_grad()
    ret = net(ex)
    if prefetch is not None: prefetch.submit(clouds)
    ret["loss"].mean().backward()
    hip_optim.clip_grad_norm_(params, 10.0, optimizer=opt)
    opt.step()
    net.update_global_step()
for _ in range(6): step()
torch.cuda.synchronize()
shapes = os.environ.get("RSLO_SITES_SHAPES") == "1"      # also print the operand shapes of every launch
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=shapes,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(); torch.cuda.synchronize()
NATIVE = ("at::native", "__amd_rocclr", "Cijk_", "igemm", "batched_transpose", "void at::")
sites = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
total = 0
for e in prof.events():
    ks = [k for k in (getattr(e, "kernels", None) or []) if k.name.startswith(NATIVE)]
    if not ks: continue
    # innermost python frame of ours
    p, site, top = e, None, e.name
    while p is not None:
        st = getattr(p, "stack", None) or []
        for fr in st:
            if "rslo_amd/" in fr and "scripts/" not in fr:
                site = fr.split("rslo_amd/", 1)[1]; break
        if site: break
        top = p.name
        p = p.cpu_parent
    if site is None:
        q = e
        while q.cpu_parent is not None: q = q.cpu_parent
        site = "<bwd> " + q.name
    s = sites[site]
    s[0] += len(ks); s[1] += sum(k.duration for k in ks); s[2][e.name + (" " + str(e.input_shapes) if shapes else "")] += len(ks)
    total += len(ks)
print("torch-native launches in one step:", total)
for site, (n, us, ops) in sorted(sites.items(), key=lambda kv: -kv[1][0]):
    print("%4d %8.1f us  %-80s %s" % (n, us, site[:80], dict(ops.most_common(12 if shapes else 4))))
if prefetch is not None: prefetch.close()
