"""Finer sync-bracketed timings inside network_forward and loss (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import workload
torch.manual_seed(7)
net, _ = workload.build_network(); net.train(); net.global_step.fill_(2000)
clouds = [[torch.from_numpy(c).cuda() for c in pair] for pair in workload.kitti_pairs(4)]
acc = {}
def T(): torch.cuda.synchronize(); return time.perf_counter()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = T(); r = f(*a, **k); acc[label] = acc.get(label, 0) + T() - t0; return r
    setattr(obj, name, w)
import rslo.data.dataset as D, rslo.models.odom_pred as OP, rslo.models.voxel_odom_net as VN
wrap(net.middle_feature_extractor, "forward", "encoder fwd")
wrap(net.odom_predictor, "vote", "head.vote")
wrap(net, "gen_tq_maps", "loss.gen_tq_maps")
wrap(net._consistency_loss, "pair_losses", "loss.consistency pair_losses")
wrap(net._translation_loss, "_compute_loss", "loss.AdaptiveL2 (T+pyramid T)")
wrap(net._rotation_loss, "_compute_loss", "loss.AdaptiveL2 (R+pyramid R)")
for blk in ("tq_map_conv", "t_map_conf", "q_map_conf"):
    wrap(getattr(net.odom_predictor, blk), "forward", "head." + blk)
for i in range(3):
    wrap(net.odom_predictor.blocks[i], "forward", "head.blocks")
    wrap(net.odom_predictor.deblocks[i], "forward", "head.deblocks")
    wrap(net.odom_predictor.skip_blocks[i], "forward", "head.skip_blocks")
wrap(net.odom_predictor, "forward", "head total fwd")
wrap(net, "create_loss", "create_loss total")
for it in range(7):
    if it == 3: acc.clear()
    ex = workload.make_example(net, clouds)
    ret = net(ex)
    t0 = T(); ret["loss"].mean().backward(); acc["backward"] = acc.get("backward", 0) + T() - t0
    net.zero_grad(set_to_none=True)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("%-36s %7.2f ms/step" % (k, 1e3 * v / 4))
