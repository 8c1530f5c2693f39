"""Golden vectors for the training-side row by IMPORTING the reference's own
torchplus.train + builders from /root/reference (authoring container only):

    python tests/golden/make_golden_train.py      ->  tests/golden/train_side.npz / train_side.json

Pinned: the (non-BN, BN) x 4-layer-group split of the optimizer builder, the OneCycle lr / beta1 trajectory from the
shipped prototxt optimizer block (real protobuf parse), decoupled weight decay + Adam(0.9->mom, 0.99) parameter
trajectories under scripted gradients, the optimizer state_dict layout, and the checkpoints.json bookkeeping
(max_to_keep eviction).  Shims: collections.Iterable alias; PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python; import-only stand-ins for
numba / tensorboardX / fire.
"""
import collections
import collections.abc
import json
import os
import sys
import tempfile

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")
for n in ("Iterable", "Mapping", "Sequence"):
    if not hasattr(collections, n):
        setattr(collections, n, getattr(collections.abc, n))

TOTAL_STEP = 40


class TinyLoss(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor([a]))


def tiny_net():
    """Same attribute names the optimizer builder walks (optimizer_builder.py:49-66), toy sizes."""
    torch.manual_seed(0)
    net = nn.Module()
    net.voxel_feature_extractor = nn.Sequential()                      # parameter-free, like the shipped VFE
    net.middle_feature_extractor = nn.Sequential(nn.Linear(4, 6), nn.BatchNorm1d(6), nn.Linear(6, 3, bias=False))
    net.odom_predictor = nn.Sequential(nn.Sequential(nn.Conv2d(2, 3, 3), nn.BatchNorm2d(3)), nn.Linear(3, 2))
    net._rotation_loss = TinyLoss(-2.5)
    net._translation_loss = TinyLoss(0.0)
    net._pyramid_rotation_loss = net._rotation_loss
    net._pyramid_translation_loss = net._translation_loss
    net._consistency_loss = TinyLoss(0.0)
    net._consistency_loss.alpha.requires_grad = False
    return net


def scripted_grads(net, step):
    for i, p in enumerate(net.parameters()):
        if p.requires_grad:
            idx = torch.arange(p.numel(), dtype=torch.float32, device=p.device).reshape(p.shape)
            p.grad = 0.1 * torch.sin(0.37 * idx + 0.11 * step + i) + 0.01 * p.detach()


def run(opt_builder, lr_builder, optimizer_cfg, net):
    opt = opt_builder.build(optimizer_cfg, net, mixed=False, loss_scale=-1.0)
    sched = lr_builder.build(optimizer_cfg, opt, TOTAL_STEP)
    lrs, moms, traj = [], [], []
    for step in range(TOTAL_STEP):
        sched.step(step)
        lrs.append([g["lr"] for g in opt.param_groups])
        moms.append([g["betas"][0] for g in opt.param_groups])
        scripted_grads(net, step)
        opt.step()
        opt.zero_grad()
        traj.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy().copy())
    return opt, np.array(lrs, np.float64), np.array(moms, np.float64), np.stack(traj)


def main():
    import types
    for name in ("numba", "tensorboardX", "fire"):      # import-only stand-ins (never executed here)
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
            m.SummaryWriter = object
            sys.modules[name] = m
    sys.path.insert(0, os.path.join(REF, "rslo"))
    sys.path.insert(0, REF)
    from google.protobuf import text_format
    from rslo.builder import lr_scheduler_builder, optimizer_builder
    from rslo.protos import pipeline_pb2
    import torchplus.train as tp_train

    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    with open(os.path.join(REF, "config", "kitti_train_ours.prototxt")) as f:
        text_format.Merge(f.read(), cfg)
    net = tiny_net()
    opt, lrs, moms, traj = run(optimizer_builder, lr_scheduler_builder, cfg.train_config.optimizer, net)
    sd = opt.state_dict()
    layout = {"group_sizes": [len(g["params"]) for g in sd["param_groups"]],
              "group_keys": sorted(k for k in sd["param_groups"][0] if k != "params"),
              "state_keys": sorted(sd["state"][0].keys()), "name": opt.name,
              "weight_decay_in_groups": [g["weight_decay"] for g in sd["param_groups"]]}

    # checkpoint bookkeeping: 5 saves with max_to_keep 3, then latest lookup
    with tempfile.TemporaryDirectory() as d:
        net.name = "voxelnet"
        for step in (10, 20, 30, 40, 50):
            tp_train.save_models(d, [net, opt], step, max_to_keep=3)
        index = json.load(open(os.path.join(d, "checkpoints.json")))
        files = sorted(f for f in os.listdir(d))
        latest = os.path.basename(tp_train.latest_checkpoint(d, "voxelnet"))
    # sampler index lists (numpy RNG seed 7 inside the reference)
    from rslo.utils import distributed_utils as DU
    data = list(range(50))
    samplers = {}
    for rank in (0, 1):
        samplers["epoch_r%d" % rank] = [[int(i), int(s)] for i, s in
                                        DU.DistributedGivenIterationSamplerEpoch(data, 60, 2, world_size=2, rank=rank)]
        samplers["epoch_resume_r%d" % rank] = [[int(i), int(s)] for i, s in DU.DistributedGivenIterationSamplerEpoch(
            data, 60, 2, world_size=2, rank=rank, last_iter=9)]
        samplers["epoch_review_r%d" % rank] = [[int(i), int(s)] for i, s in DU.DistributedGivenIterationSamplerEpoch(
            data, 60, 2, world_size=2, rank=rank, review_cycle=0.5)]
        samplers["given_r%d" % rank] = [int(i) for i in DU.DistributedGivenIterationSampler(data, 60, 2, world_size=2,
                                                                                              rank=rank)]
        samplers["seq_r%d" % rank] = [int(i) for i in DU.DistributedSequatialSampler(list(range(51)), 2, rank)]
    json.dump(samplers, open(os.path.join(HERE, "samplers.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "train_side.npz"), lrs=lrs, moms=moms, traj=traj.astype(np.float32))
    json.dump({"layout": layout, "ckpt_index": index, "ckpt_files": files, "ckpt_latest": latest,
               "total_step": TOTAL_STEP}, open(os.path.join(HERE, "train_side.json"), "w"), indent=1)
    print("lr first/peak/last", lrs[0], lrs.max(0), lrs[-1])
    print(layout, files, latest)


if __name__ == "__main__":
    main()
