"""The structure plan of one C3 batch (8 clouds: voxelization + site hashes + every rulebook + pair lists + row orders, one
foreign call: rslo_plan_encoder) ALONE on the GPU: ms per plan.  Under rocprofv3 --stats: the chain kernel by kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, rslo_amd
from rslo_amd import workload
from rslo_amd.plan import EncoderPlanner
torch.manual_seed(7)
net, _ = workload.build_network()
net.train()
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
pl = EncoderPlanner(net, 40000)
for _ in range(3):
    pl.submit(clouds)
torch.cuda.synchronize()
n = int(os.environ.get("N", "50"))
t0 = time.perf_counter()
for _ in range(n):
    pl.submit(clouds)
torch.cuda.synchronize()
print("plan alone: %.3f ms per batch of 8 clouds" % (1e3 * (time.perf_counter() - t0) / n))
