"""Where a C2 inference step goes: graph replays alone, structure plans alone (1 / 2 streams), both (rslo_amd/inference.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, rslo_amd
import spconv
from rslo.models import middle
from rslo_amd import inference, synthetic as S
gen = spconv.utils.VoxelGenerator(list(S.VOXEL_SIZE), list(S.PC_RANGE), S.MAX_POINTS_PER_VOXEL, S.MAX_VOXELS)
torch.manual_seed(7)
enc = middle.get_middle_class("SpMiddleFHDWithCov2_3")([1] + gen.grid_size[::-1].tolist() + [7], bn_type="None", use_leakyReLU=True,
                                                       num_input_features=7, num_filters_down1=[], num_filters_down2=[]).cuda().eval()
class N: middle_feature_extractor, voxel_generator, training = enc, gen, False
r = inference.EncoderGraphRunner(N(), S.MAX_VOXELS, torch.device("cuda", 0))
c = torch.from_numpy(S.scan(n_el=64, scan_seed=1)).cuda()
hs = [r.submit(c) for _ in range(4)]
for h in hs: r.run(h)
torch.cuda.synchronize()
g = r._graphs[0][0]
def t(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("graph replay alone: %.3f ms" % t(g.replay))
pl = r.planner
def plan_on(stream, slot):
    with torch.cuda.stream(stream):
        pl.submit([[c]], with_pairs=False, slot=slot, point_capacity=r.point_capacity)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
print("plan alone, one stream: %.3f ms" % t(lambda: plan_on(s0, 1)))
k = [0]
def two():
    k[0] += 1
    plan_on(s0 if k[0] & 1 else s1, 1 + (k[0] & 1))
print("plan alone, two streams alternating: %.3f ms" % t(two))
def both():
    k[0] += 1
    plan_on(s0 if k[0] & 1 else s1, 1 + (k[0] & 1))
    g.replay()
print("plan (two streams) + replay of arena 0's graph from one thread: %.3f ms" % t(both))
t0 = time.perf_counter()
for _ in range(300): pl.submit([[c]], with_pairs=False, slot=1, point_capacity=r.point_capacity)
print("host time of planner.submit: %.3f ms (GPU not waited for)" % (1e3 * (time.perf_counter() - t0) / 300)); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): g.replay()
print("host time of graph.replay: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 300)); torch.cuda.synchronize()
r.close()
