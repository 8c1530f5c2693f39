"""Who launches kernels whose name matches PATTERN (argv[1])?  Prints (module label, aten op chain) counts."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from torch.profiler import profile, ProfilerActivity, record_function
from rslo_amd import workload
pats = sys.argv[1:]
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
for name, m in net.named_modules():
    if name and name.count(".") < 4:
        st = []
        m.register_forward_pre_hook(lambda mod, a, n=name, st=st: st.append(record_function("M:" + n).__enter__()) or None)
        m.register_forward_hook(lambda mod, a, o, st=st: st.pop().__exit__(None, None, None) or None)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=8e-5, fused=True)
def step():
    ex = workload.make_example(net, clouds)
    opt.zero_grad(set_to_none=True)
    ret = net(ex); ret["loss"].mean().backward()
    with record_function("F:clip"): torch.nn.utils.clip_grad_norm_(params, 10.0)
    with record_function("F:adam"): opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
for pat in pats:
    cnt = collections.Counter()
    for e in prof.events():
        ks = [k for k in (getattr(e, "kernels", None) or []) if pat in k.name]
        if not ks: continue
        chain, p = [], e
        while p is not None and len(chain) < 6:
            chain.append(p.name[:48]); p = p.cpu_parent
        cnt[" <- ".join(chain)] += len(ks)
    print("==", pat)
    for k, v in cnt.most_common(12): print("%4d  %s" % (v, k))
