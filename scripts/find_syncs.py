"""List every host<->device synchronisation point of one C3 step (torch sync-debug mode + stack)."""
import os, sys, traceback, warnings, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=8e-5, fused=True)
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
pf = workload.ExamplePrefetcher(net); pf.submit(clouds)
def step():
    ex = pf.get()
    opt.zero_grad(set_to_none=True)
    ret = net(ex); pf.submit(clouds); ret["loss"].mean().backward()
    torch.nn.utils.clip_grad_norm_(params, 10.0); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
sites = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/rslo_amd/" in f.filename or "scripts/" in f.filename]
    sites[" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-4:]))] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
step()
torch.cuda.set_sync_debug_mode(0)
for k, v in sites.most_common(): print(v, k)
