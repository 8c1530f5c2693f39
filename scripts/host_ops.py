"""Host-side cost of one C3 step by operator (torch.profiler, CPU self time): which ops the issuing threads spend time in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from torch.profiler import profile, ProfilerActivity
from rslo_amd import workload
from rslo.builder import lr_scheduler_builder, optimizer_builder
from rslo.utils import config_text
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
cfg = config_text.shipped_config().train_config
opt = optimizer_builder.build(cfg.optimizer, net)
params = [p for p in net.parameters() if p.requires_grad]
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
ex = workload.make_example(net, clouds); net.plan_example(ex)
def step():
    opt.zero_grad()
    ret = net(dict(ex))
    ret["loss"].mean().backward()
    torch.nn.utils.clip_grad_norm_(params, 10.0); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)[:45]
tot = sum(e.self_cpu_time_total for e in prof.key_averages())
print("total self CPU per step: %.2f ms" % (tot / 5 / 1e3))
for e in rows:
    print("%8.3f ms/step  %5d calls/step  %s" % (e.self_cpu_time_total / 5 / 1e3, e.count // 5, e.key[:90]))
