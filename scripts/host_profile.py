"""cProfile of the host side of a C3 step (where does Python/launch time go?)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=8e-5, fused=True)
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
def step():
    ex = workload.make_example(net, clouds)
    opt.zero_grad(set_to_none=True)
    ret = net(ex); ret["loss"].mean().backward()
    torch.nn.utils.clip_grad_norm_(params, 10.0); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
