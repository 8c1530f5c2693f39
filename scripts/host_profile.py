"""cProfile of the issuing thread over a few C3 steps (host-side cost centres).  python scripts/host_profile.py [steps]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import workload
from rslo.builder import lr_scheduler_builder, optimizer_builder
from rslo.utils import config_text
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if os.environ.get('ONE_THREAD', '1') == '1':      # backward on the calling thread: its cost centres show up in the same profile
    torch.autograd.set_multithreading_enabled(False)
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
cfg = config_text.shipped_config().train_config
opt = optimizer_builder.build(cfg.optimizer, net)
sched = lr_scheduler_builder.build(cfg.optimizer, opt, cfg.steps)
params = [p for p in net.parameters() if p.requires_grad]
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
pf = workload.ExamplePrefetcher(net); pf.submit(clouds)
def step():
    ex = pf.get()
    sched.step(net.get_global_step()); opt.zero_grad()
    ret = net(ex)
    pf.submit(clouds)
    ret["loss"].mean().backward()
    torch.nn.utils.clip_grad_norm_(params, 10.0); opt.step(); net.update_global_step()
for _ in range(8): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime")
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(70)
out = buf.getvalue()
print("\n".join(l[:150] for l in out.splitlines()))
