"""cProfile of what the prefetch thread does per step (make_example + plan_example), run in the main thread."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train()
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
def job():
    ex = workload.make_example(net, clouds)
    net.plan_example(ex)
    return ex
for _ in range(5): job()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): job()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime")
buf = io.StringIO(); st.stream = buf; st.print_stats(32)
print("\n".join(l[:160] for l in buf.getvalue().splitlines()))
