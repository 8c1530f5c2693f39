"""B0: the drivers' own step sequences against the mirror.

`train_hdf5.py:455-480` (setup) and `:600-690` (the step) and `evaluate.py:363-420` (the evaluation loop) are replayed
statement by statement -- same calls, same order, the same objects the drivers construct (apex.amp.initialize,
apex.parallel.DistributedDataParallel, the builders, torchplus.train checkpoints) -- (a) on the scripted toy problem whose
parameter trajectory the REFERENCE's own optimizer modules produced (tests/golden/train_side.npz, made by
tests/golden/make_golden_train.py), with a checkpoint save / restore in the middle; (b) on the shipped network on the GPU.
Nothing here reads /root/reference."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

import rslo_amd  # noqa: F401

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from make_golden_train import TOTAL_STEP, tiny_net   # noqa: E402  (the scripted toy problem, not reference code)


def _shipped_train_cfg():
    from rslo.utils import config_text
    return config_text.shipped_config().train_config


class _SingleRankGroup:
    """train_hdf5.py runs with --use_dist True even on one GPU: a world-size-1 process group."""

    def __init__(self, backend):
        self.backend = backend

    def __enter__(self):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group(self.backend, rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)

    def __exit__(self, *exc):
        dist.destroy_process_group()
        return False


def _toy_driver_net(device):
    """tiny_net() + what the driver touches on a network (train_hdf5.py:410,582,611,623,674): forward(example) -> dict with
    "loss", the global step counter.  The loss is linear in the parameters with the scripted gradient of
    make_golden_train.scripted_grads as coefficient, so backward() reproduces that gradient exactly."""
    net = tiny_net().to(device)
    net.name = "voxelnet"
    net.register_buffer("global_step", torch.zeros(1, dtype=torch.int64, device=device))

    def forward(example):
        step = int(example["step"])
        loss = 0.0
        for i, p in enumerate(net.parameters()):
            if p.requires_grad:
                idx = torch.arange(p.numel(), dtype=torch.float32, device=p.device).reshape(p.shape)
                c = 0.1 * torch.sin(0.37 * idx + 0.11 * step + i) + 0.01 * p.detach()
                loss = loss + (p * c).sum()
        z = torch.zeros(1, device=device)
        return {"loss": loss.reshape(1), "translation_loss": z, "rotation_loss": z, "pyramid_loss": z}
    net.forward = forward
    net.get_global_step = lambda: int(net.global_step.item())
    net.update_global_step = lambda: net.global_step.add_(1)
    return net


def _driver_setup(net, model_dir, opt_level):
    """train_hdf5.py:408-480 in the driver's order: optimizer, restore, amp, DDP wrapper, schedule."""
    import apex
    import torchplus
    from rslo.builder import lr_scheduler_builder, optimizer_builder
    cfg = _shipped_train_cfg()
    fastai_optimizer = optimizer_builder.build(cfg.optimizer, net, mixed=False, loss_scale=-1.0)      # :408-411
    torchplus.train.try_restore_latest_checkpoints(model_dir, [net])                                  # :449
    torchplus.train.try_restore_latest_checkpoints(model_dir, [fastai_optimizer])                     # :450-451
    apex.amp.register_float_function(torch, "svd")                                                    # :457-458
    apex.amp.register_float_function(torch, "matmul")
    net, amp_optimizer = apex.amp.initialize(net, fastai_optimizer, opt_level=opt_level, keep_batchnorm_fp32=None,
                                             loss_scale=None)                                         # :460-461
    net_parallel = apex.parallel.DistributedDataParallel(net)                                         # :463
    return net, net_parallel, amp_optimizer


def _driver_step(net, net_parallel, amp_optimizer, lr_scheduler, example, world=1):
    """train_hdf5.py:618-674, statement order kept (logging reductions included)."""
    import apex
    lr_scheduler.step(net.get_global_step())                              # :618
    ret_dict = net_parallel(example)                                      # :623
    loss = ret_dict["loss"].mean()                                        # :625
    translation_loss = ret_dict["translation_loss"].mean() / world        # :626-631
    rotation_loss = ret_dict["rotation_loss"].mean() / world
    pyramid_loss = ret_dict["pyramid_loss"].mean() / world
    dev = loss.device
    consistency_loss = ret_dict.get("C_loss", torch.zeros(1, device=dev)).mean() / world
    reduced = [loss.data.clone(), translation_loss.data.clone(), rotation_loss.data.clone(), pyramid_loss.data.clone(),
               consistency_loss.data.clone()]                             # :644-649
    for t in reduced:                                                     # :650-657 (use_dist)
        dist.all_reduce(t)
    amp_optimizer.zero_grad()                                             # :661
    with apex.amp.scale_loss(loss, amp_optimizer) as scaled_loss:         # :663-664
        scaled_loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)                # :671
    amp_optimizer.step()                                                  # :672
    net.update_global_step()                                              # :674
    return ret_dict, reduced


def _toy_trajectory(device, opt_level, tmp_path, backend):
    import apex
    import torchplus
    from rslo.builder import lr_scheduler_builder
    cfg = _shipped_train_cfg()
    model_dir = str(tmp_path)
    traj = []
    with _SingleRankGroup(backend):
        try:
            net = _toy_driver_net(device)
            net, net_parallel, amp_optimizer = _driver_setup(net, model_dir, opt_level)
            assert net_parallel.module is net
            lr_scheduler = lr_scheduler_builder.build(cfg.optimizer, amp_optimizer, TOTAL_STEP)       # :478-480
            for step in range(TOTAL_STEP):
                if step == TOTAL_STEP // 2:
                    # :734 save_models([net, amp_optimizer]) ... a new process resumes: :449-451 restore both, same schedule
                    torchplus.train.save_models(model_dir, [net, amp_optimizer], net.get_global_step())
                    net = _toy_driver_net(device)
                    with torch.no_grad():
                        for p in net.parameters():
                            p.add_(1.0)                   # a resumed process starts from other values: the restore must win
                    net, net_parallel, amp_optimizer = _driver_setup(net, model_dir, opt_level)
                    assert net.get_global_step() == step
                    lr_scheduler = lr_scheduler_builder.build(cfg.optimizer, amp_optimizer, TOTAL_STEP)
                _driver_step(net, net_parallel, amp_optimizer, lr_scheduler, {"step": net.get_global_step()})
                traj.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy().copy())
        finally:
            apex.amp.initialize(net, opt_level="O0")      # the precision mode is process-wide
    return np.stack(traj)


@pytest.mark.parametrize("opt_level", ["O0", "O1"])
def test_driver_step_order_reproduces_the_reference_trajectory_on_the_host(tmp_path, opt_level):
    g = np.load(os.path.join(GOLD, "train_side.npz"))
    traj = _toy_trajectory("cpu", opt_level, tmp_path, "gloo")
    np.testing.assert_allclose(traj, g["traj"], rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("opt_level", ["O0", "O1"])
def test_driver_step_order_reproduces_the_reference_trajectory_on_the_gpu(tmp_path, opt_level, hip):
    """The same replay on cuda:0: the multi-tensor clip + Adam of optim.hip behind OptimWrapper, RCCL group of one rank,
    checkpoints written from and restored to device tensors."""
    g = np.load(os.path.join(GOLD, "train_side.npz"))
    traj = _toy_trajectory("cuda", opt_level, tmp_path, "nccl")
    np.testing.assert_allclose(traj, g["traj"], rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_driver_step_and_eval_loop_on_the_shipped_network(tmp_path, hip):
    """train_hdf5.py:455-480, 600-690 and evaluate.py:363-420 against the shipped network on the GPU: three training steps
    through the driver's objects, a checkpoint after step 2 restored into a freshly built network whose third step must
    reproduce the uninterrupted run's; then the evaluation loop (no_grad forward through the wrapper, [n,7] results,
    all_gather, reshape)."""
    import apex
    import torchplus
    from rslo.builder import lr_scheduler_builder
    from rslo_amd import synthetic, workload
    cfg = _shipped_train_cfg()
    model_dir = str(tmp_path)
    torch.cuda.set_device(0)
    pairs = []
    for i in range(2):
        p0, p1, _ = synthetic.frame_pair(i)
        pairs.append([p0[::5].copy(), p1[::5].copy()])

    def build():
        torch.manual_seed(7)
        net, _ = workload.build_network()
        net.train()
        return net

    def example(net, i):
        return workload.make_example(net, [pairs[i % 2]])

    def snapshot(net):
        return torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()

    with _SingleRankGroup("nccl"):
        try:
            net = build()
            net.global_step.fill_(1999)          # crosses train_hdf5's icp warm-up boundary (identity pose up to step 1500)
            net, net_parallel, amp_optimizer = _driver_setup(net, model_dir, "O0")
            lr_scheduler = lr_scheduler_builder.build(cfg.optimizer, amp_optimizer, cfg.steps)
            losses = []
            for i in range(3):
                if i == 2:
                    torchplus.train.save_models(model_dir, [net, amp_optimizer], net.get_global_step())
                    before = snapshot(net)
                ret, reduced = _driver_step(net, net_parallel, amp_optimizer, lr_scheduler, example(net, i))
                assert torch.isfinite(ret["loss"]).all()
                losses.append(float(reduced[0]))
                for k in ("loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss", "translation_preds",
                          "rotation_preds"):
                    assert k in ret, k
            assert net.get_global_step() == 2002
            after = snapshot(net)
            assert not torch.equal(before, after)

            # a new process: fresh network + optimizer, both restored (:449-451), then the same third step
            net2 = build()
            net2, net_parallel2, amp_optimizer2 = _driver_setup(net2, model_dir, "O0")
            assert net2.get_global_step() == 2001 and torch.equal(snapshot(net2), before)
            lr_scheduler2 = lr_scheduler_builder.build(cfg.optimizer, amp_optimizer2, cfg.steps)
            ret2, reduced2 = _driver_step(net2, net_parallel2, amp_optimizer2, lr_scheduler2, example(net2, 2))
            assert abs(float(reduced2[0]) - losses[2]) <= 1e-4 * abs(losses[2])
            # agreement of the UPDATE, tensor by tensor (Adam's first steps move every weight by ~lr, so a lost optimizer
            # state shows as O(1) everywhere).  Tensors whose gradient is rounding noise -- conv biases in front of a
            # BatchNorm, the softmax-shift bias of a confidence head -- get full-size Adam steps of arbitrary sign in ANY run,
            # hence the statistic: median over tensors tiny, nine tenths of them within 2 %.
            rel_upd = []
            snap2 = snapshot(net2)
            off = 0
            for p_ in net.parameters():
                n_ = p_.numel()
                u1 = (after[off:off + n_] - before[off:off + n_]).double()
                u2 = (snap2[off:off + n_] - before[off:off + n_]).double()
                off += n_
                if float(u1.norm()) > 0:
                    rel_upd.append(float((u2 - u1).norm() / u1.norm()))
            rel_upd = np.array(rel_upd)
            assert len(rel_upd) >= 200 and np.median(rel_upd) < 1e-3 and np.mean(rel_upd < 2e-2) > 0.9, (
                np.median(rel_upd), np.mean(rel_upd < 2e-2))

            # evaluate.py:363-420
            net2.eval()
            results = []
            for i in range(2):
                with torch.no_grad():
                    pred = net_parallel2(example(net2, i))
                    assert "voxel_features" in pred or not getattr(net2, "testing", False)
                    results.append(torch.cat([pred["translation_preds"], pred["rotation_preds"]], dim=1))
            results = torch.cat(results, dim=0)
            dist.barrier()
            gather_list = [torch.zeros_like(results) for _ in range(dist.get_world_size())]
            dist.all_gather(gather_list, results)
            results = torch.cat(gather_list, dim=-1).reshape([-1, 7])[:2]
            assert results.shape == (2, 7) and torch.isfinite(results).all()
            assert torch.allclose(results[:, 3:].norm(dim=1), torch.ones(2, device=results.device), atol=1e-5)
            net2.train()
        finally:
            apex.amp.initialize(net, opt_level="O0")
