"""N > 1 path with world_size-2 gloo on CPU: index sharding and the gradient all-reduce (the only exchange step
of the hot path besides SyncBN statistics).  The dense BEV head is plain torch, so it runs on CPU here."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rslo_amd  # noqa: F401

PC_RANGE = np.array([-70.4, -38.4, -3, 70.4, 38.4, 5], np.float32)


def small_head(bn_type="BN"):
    import rslo.models.odom_pred as OP
    return OP.get_odom_class("UNRResNetOdomPredEncDecSVDTempMask")(
        bn_type=bn_type, enc_use_norm=True, conv_type="mask_conv", layer_nums=[1, 1, 1], layer_strides=[2, 2, 2],
        num_filters=[8, 8, 16], upsample_strides=[2, 2, 2], num_upsample_filters=[8, 8, 8], num_input_features=8,
        pooling_type="avg_pool", pooling_size=1, dropout=1e-22, cycle_constraint=True, pred_pyramid_motion=True,
        use_deep_supervision=True, odom_format="rx+t", point_cloud_range=PC_RANGE, dense_predict=True,
        conf_type="softmax", use_svd=False, cubic_pred_height=0)


def rank_input(rank):
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(1, 4, 16, 24, generator=g) * (torch.rand(1, 1, 16, 24, generator=g) > 0.5) for _ in range(2)]
    return xs


def head_loss(head, xs):
    out = head([x.clone() for x in xs])
    return (out["translation_preds"][0] ** 2).sum() + (out["rotation_preds"][0][:, 1:] ** 2).sum()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rslo.utils.distributed_utils import (DistributedGivenIterationSamplerEpoch, average_gradients,
                                              broadcast_params)
    torch.manual_seed(rank)              # different init per rank on purpose
    head = small_head()
    broadcast_params(head, 0)            # ... made identical by the broadcast
    head.train()
    # (1) manual bucketed all-reduce
    head_loss(head, rank_input(rank)).backward()
    average_gradients(head)
    g_manual = torch.cat([p.grad.reshape(-1) for p in head.parameters() if p.grad is not None])
    # (2) DDP with unused parameters tolerated (77 of 290 tensors never get a gradient in the real net)
    head2 = small_head()
    head2.load_state_dict(head.state_dict())
    ddp = torch.nn.parallel.DistributedDataParallel(head2, find_unused_parameters=True)
    out = ddp([x.clone() for x in rank_input(rank)])
    ((out["translation_preds"][0] ** 2).sum() + (out["rotation_preds"][0][:, 1:] ** 2).sum()).backward()
    g_ddp = torch.cat([p.grad.reshape(-1) for p in head2.parameters() if p.grad is not None])
    sampler = DistributedGivenIterationSamplerEpoch(dataset_len=50, total_iter=6, batch_size=2)
    q.put((rank, g_manual.numpy(), g_ddp.numpy(), [i for i, _ in sampler], {k: v.numpy() for k, v in head.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gradient_allreduce_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, gm0, gd0, idx0, sd0), (_, gm1, gd1, idx1, sd1) = res
    # both ranks end with the same averaged gradient; manual bucket == DDP
    np.testing.assert_allclose(gm0, gm1, rtol=0, atol=0)
    np.testing.assert_allclose(gm0, gd0, rtol=1e-5, atol=1e-7)
    # equals the mean of the two per-rank gradients computed in one process
    head = small_head()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    head.train()
    grads = []
    for r in range(world):
        head.zero_grad()
        h = small_head()
        h.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})   # fresh BN running stats per rank
        h.train()
        head_loss(h, rank_input(r)).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in h.parameters() if p.grad is not None]).numpy())
    np.testing.assert_allclose(gm0, (grads[0] + grads[1]) / 2, rtol=1e-4, atol=1e-6)
    # sharding: same global shuffle, disjoint contiguous slices, 6 iterations x batch 2 each
    assert len(idx0) == len(idx1) == 12
    g = torch.Generator().manual_seed(7)
    perm = torch.cat([torch.randperm(50, generator=g) for _ in range(1)])[:24].tolist()
    assert idx0 == perm[:12] and idx1 == perm[12:24]


def test_sampler_resume():
    from rslo.utils.distributed_utils import DistributedGivenIterationSamplerEpoch as S
    full = [i for i, _ in S(100, 10, 3, world_size=4, rank=2)]
    resumed = [i for i, _ in S(100, 10, 3, world_size=4, rank=2, last_iter=3)]
    assert resumed == full[12:] and len(full) == 30
