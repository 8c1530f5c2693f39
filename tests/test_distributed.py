"""N > 1 path with world_size-2 gloo on CPU: index sharding and the gradient all-reduce (the only exchange step
of the hot path besides SyncBN statistics).  The dense BEV head is plain torch, so it runs on CPU here."""
import os
import socket

import numpy as np
import pytest
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
    from rslo_amd import streams
    assert streams.SHARE_SIDE == "auto" and not streams.sharing()      # a single process keeps its two side streams apart
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a rank of a data-parallel job issues the dense weight gradients on the covariance branch's stream: the step tolerates four
    # active streams and the collective library's is one of them (rslo_amd/streams.py, profiles/r06_fifth_stream.txt)
    assert streams.sharing()
    from rslo.utils.distributed_utils import (DistributedGivenIterationSamplerEpoch, average_gradients,
                                              broadcast_params)
    torch.manual_seed(rank)              # different init per rank on purpose
    head = small_head()
    broadcast_params(head, 0)            # ... made identical by the broadcast
    head.train()
    # (1) manual bucketed all-reduce
    head_loss(head, rank_input(rank)).backward()
    average_gradients(head, mean=True)
    g_manual = torch.cat([p.grad.reshape(-1) for p in head.parameters() if p.grad is not None])
    # (2) DDP with unused parameters tolerated (77 of 290 tensors never get a gradient in the real net)
    head2 = small_head()
    head2.load_state_dict(head.state_dict())
    ddp = torch.nn.parallel.DistributedDataParallel(head2, find_unused_parameters=True)
    out = ddp([x.clone() for x in rank_input(rank)])
    ((out["translation_preds"][0] ** 2).sum() + (out["rotation_preds"][0][:, 1:] ** 2).sum()).backward()
    g_ddp = torch.cat([p.grad.reshape(-1) for p in head2.parameters() if p.grad is not None])
    sampler = DistributedGivenIterationSamplerEpoch(list(range(50)), total_iter=60, batch_size=2)
    q.put((rank, g_manual.numpy(), g_ddp.numpy(), [[int(i), int(sd)] for i, sd in sampler], {k: v.numpy() for k, v in head.state_dict().items()}))
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
    # sharding: the reference's own index lists (tests/golden/samplers.json), rank/world taken from the process group
    gold = samplers_gold()
    assert idx0 == gold["epoch_r0"] and idx1 == gold["epoch_r1"]


def samplers_gold():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "samplers.json")))


def test_samplers_match_reference_index_lists():
    """Index lists produced by the reference's samplers (make_golden_train.py): epoch-wise with seeds, resume,
    review cycle, single-shuffle variant and the sequential eval shard."""
    from rslo.utils import distributed_utils as DU
    gold = samplers_gold()
    data = list(range(50))
    for rank in (0, 1):
        as_list = lambda it: [[int(i), int(s)] for i, s in it]   # noqa: E731
        assert as_list(DU.DistributedGivenIterationSamplerEpoch(data, 60, 2, world_size=2, rank=rank)) == gold["epoch_r%d" % rank]
        assert as_list(DU.DistributedGivenIterationSamplerEpoch(data, 60, 2, world_size=2, rank=rank, last_iter=9)) \
            == gold["epoch_resume_r%d" % rank]
        assert as_list(DU.DistributedGivenIterationSamplerEpoch(data, 60, 2, world_size=2, rank=rank, review_cycle=0.5)) \
            == gold["epoch_review_r%d" % rank]
        assert [int(i) for i in DU.DistributedGivenIterationSampler(data, 60, 2, world_size=2, rank=rank)] \
            == gold["given_r%d" % rank]
        assert [int(i) for i in DU.DistributedSequatialSampler(list(range(51)), 2, rank)] == gold["seq_r%d" % rank]
    s = DU.DistributedGivenIterationSamplerEpoch(data, 60, 2, world_size=2, rank=1, last_iter=9)
    assert len(s) == 120 and len(list(s)) == 100


def test_average_gradients_sums_like_the_reference_single_process():
    """World size 1 / no process group: no-op; gradients_multiply scales in place."""
    from rslo.utils.distributed_utils import average_gradients, gradients_multiply
    lin = torch.nn.Linear(3, 2)
    lin(torch.ones(1, 3)).sum().backward()
    g = lin.weight.grad.clone()
    average_gradients(lin)
    assert torch.equal(lin.weight.grad, g)
    gradients_multiply(lin, 0.5)
    assert torch.equal(lin.weight.grad, g * 0.5)


class _TwoBranch(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b, self.never = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)

    def forward(self, x, use_b):
        y = self.a(x)
        return y + self.b(x) if use_b else y


def _uneven_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rslo.utils.distributed_utils import average_gradients
    torch.manual_seed(0)
    net = _TwoBranch()
    x = torch.ones(1, 3) * (rank + 1)
    out = {}
    # step 1: both ranks use branch b -> agreed set {a, b}; `never` stays without a gradient on every rank
    net(x, True).sum().backward()
    average_gradients(net)
    out["s1_b"] = net.b.weight.grad.clone().numpy()
    assert net.never.weight.grad is None
    # step 2: rank 1's graph skips branch b (set_to_none zero_grad): the bucket keeps its size, rank 1 sends zeros
    net.zero_grad(set_to_none=True)
    net(x, rank == 0).sum().backward()
    average_gradients(net)
    out["s2_a"] = net.a.weight.grad.clone().numpy()
    out["s2_b"] = net.b.weight.grad.clone().numpy()
    # step 3: a parameter outside the agreed set receives its FIRST gradient, on one rank only: both ranks re-agree in
    # the same call (one flag over the host group) and reduce it -- no raise, no hang, nothing dropped
    net.zero_grad(set_to_none=True)
    (net(x, True).sum() + (net.never(x).sum() if rank == 0 else 0.0)).backward()
    average_gradients(net)
    out["s3_never"] = net.never.weight.grad.clone().numpy()
    out["s3_a"] = net.a.weight.grad.clone().numpy()
    # step 4: the set is sticky -- `never` stays in the bucket although no rank uses it now (zeros, not None)
    net.zero_grad(set_to_none=True)
    net(x, True).sum().backward()
    average_gradients(net)
    out["s4_never"] = net.never.weight.grad.clone().numpy()
    q.put((rank, out))
    dist.destroy_process_group()


def test_gradient_bucket_keeps_its_size_when_one_rank_skips_a_branch():
    """with zero_grad(set_to_none=True) the non-None gradient set comes from each step's graph; a rank
    whose graph skips a parameter must not shrink the flat bucket (hang) -- it contributes zeros."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    r0, r1 = res[0], res[1]
    np.testing.assert_array_equal(r0["s1_b"], r1["s1_b"])
    np.testing.assert_allclose(r0["s1_b"], np.full((2, 3), 3.0))         # d/dW of sum(Wx) = x: 1 + 2
    np.testing.assert_array_equal(r0["s2_a"], r1["s2_a"])
    np.testing.assert_allclose(r0["s2_a"], np.full((2, 3), 3.0))
    np.testing.assert_array_equal(r0["s2_b"], r1["s2_b"])
    np.testing.assert_allclose(r0["s2_b"], np.full((2, 3), 1.0))         # only rank 0 (x = 1) used the branch
    np.testing.assert_array_equal(r0["s3_never"], r1["s3_never"])
    np.testing.assert_allclose(r0["s3_never"], np.full((2, 3), 1.0))     # rank 0's gradient (x = 1) + rank 1's zeros
    np.testing.assert_allclose(r0["s3_a"], np.full((2, 3), 3.0))
    np.testing.assert_allclose(r1["s4_never"], np.zeros((2, 3)))


def _overlap_worker(rank, world, port, q, rank_mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    from rslo.utils.distributed_utils import OverlappedGradientExchange, average_gradients
    torch.manual_seed(3)
    # "encoder" -> "head": the head's backward finishes first, the hook fires while the encoder's is still to run
    net = torch.nn.Module()
    net.encoder = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8))
    net.head = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
    net.unused = torch.nn.Linear(3, 3)                      # never receives a gradient (like 77 tensors of the model)
    net.forward = lambda x: net.head(net.encoder(x))
    ref = copy.deepcopy(net)
    ref.forward = lambda x: ref.head(ref.encoder(x))
    ex = OverlappedGradientExchange(net, net.head, mean=True, module_hook=(rank_mode == "module"))
    if rank_mode == "tensor":       # the boundary marked on the tensor the head consumes (what the network does)
        net.forward = lambda x: net.head(ex.watch(net.encoder(x)))
    out = []
    for step in range(3):
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 * step + rank))
        for m in (net, ref):
            m.zero_grad(set_to_none=True)
            m.forward(x).square().sum().backward()
        fired = ex.pending is not None
        ex.finish()
        average_gradients(ref, mean=True)
        same = all((a.grad is None and b.grad is None) or torch.equal(a.grad, b.grad)
                   for a, b in zip(net.parameters(), ref.parameters()))
        out.append((fired, same, ex.pending is None))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["module", "tensor"])
def test_overlapped_gradient_exchange_equals_average_gradients(mode):
    """Two gloo ranks: the head's bucket leaves from the backward hook (asynchronously, while the encoder's backward is
    still to run), the rest at finish(); every gradient equals average_gradients' bit for bit, step after step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out in res:
        assert all(same and cleared for _, same, cleared in out), out
        assert [fired for fired, _, _ in out] == [False, True, True], out     # step 0 agrees on the set, then it overlaps


def _asym_overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    from rslo.utils.distributed_utils import OverlappedGradientExchange, average_gradients
    torch.manual_seed(5)
    net = torch.nn.Module()
    net.encoder = torch.nn.Linear(6, 8)
    net.head = torch.nn.Module()
    net.head.main, net.head.side = torch.nn.Linear(8, 3), torch.nn.Linear(8, 3)
    ref = copy.deepcopy(net)
    ex = OverlappedGradientExchange(net, net.head, mean=True, module_hook=False)

    def fwd(m, x, mode, watch):
        f = m.encoder(x)
        if mode == "no_head":                  # this rank's loss never reaches the watched tensor: its hook does not fire
            if watch:
                ex.watch(m.encoder(x))
            return f.square().sum()
        if watch:
            f = ex.watch(f)
        y = m.head.main(f)
        if mode == "both":
            y = y + m.head.side(f)
        return y.square().sum()

    # step 0 agrees on the set (both branches everywhere); step 1: rank 1 skips head.side (rank 0 sent
    # [early, rest], rank 1 one bucket); step 2: rank 1's loss skips the head altogether; step 3: everything again
    plan = [("both", "both"), ("both", "main"), ("both", "no_head"), ("both", "both")]
    out = []
    for step, modes in enumerate(plan):
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 * step + rank))
        for m in (net, ref):
            m.zero_grad(set_to_none=True)
            fwd(m, x, modes[rank], m is net).backward()
        fired = ex.pending is not None
        ex.finish()
        average_gradients(ref, mean=True)
        same = all((a.grad is None and b.grad is None) or (a.grad is not None and b.grad is not None
                                                           and torch.equal(a.grad, b.grad))
                   for a, b in zip(net.parameters(), ref.parameters()))
        out.append((fired, same, float(net.head.side.weight.grad.abs().sum())))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_is_rank_symmetric_when_one_rank_skips_part_of_the_early_module():
    """the early bucket's launch must not depend on one rank's graph.  A rank that lacks a gradient of
    the early module (or whose backward never reaches it) still takes part in the same two collectives with zeros;
    results equal average_gradients on both ranks and the two ranks hold identical gradients."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_asym_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(same for _, same, _ in out), (rank, out)
    # the hook fired wherever the loss reached the head; rank 1 joined the early bucket from finish() at step 2
    assert [f for f, _, _ in res[0][1]] == [False, True, True, True]
    assert [f for f, _, _ in res[1][1]] == [False, True, False, True]
    assert [g for _, _, g in res[0][1]] == [g for _, _, g in res[1][1]]


def _late_grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    from rslo.utils.distributed_utils import OverlappedGradientExchange, average_gradients
    torch.manual_seed(5)
    net = torch.nn.Module()
    net.encoder = torch.nn.Linear(6, 8)
    net.head = torch.nn.Module()
    net.head.main, net.head.side = torch.nn.Linear(8, 3), torch.nn.Linear(8, 8)
    ref = copy.deepcopy(net)
    ex = OverlappedGradientExchange(net, net.head, mean=True, module_hook=False)

    def fwd(m, x, mode, watch):
        # "late": head.side is applied UPSTREAM of the watched tensor too (only there on this rank) -> its gradient is
        # produced after the hook has fired and the early bucket (with a zeros stand-in for it) has left
        h = m.encoder(x)
        if mode == "late":
            h = m.head.side(h)
        f = ex.watch(h) if watch else h
        y = m.head.main(f)
        if mode == "both":
            y = y + m.head.side(f)[:, :3]
        return y.square().sum()

    plan = [("both", "both"), ("both", "late"), ("both", "both")]
    out = []
    for step, modes in enumerate(plan):
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 * step + rank))
        for m in (net, ref):
            m.zero_grad(set_to_none=True)
            fwd(m, x, modes[rank], m is net).backward()
        fired = ex.pending is not None
        ex.finish()
        average_gradients(ref, mean=True)
        same = all((a.grad is None and b.grad is None) or (a.grad is not None and b.grad is not None
                                                           and torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-7))
                   for a, b in zip(net.parameters(), ref.parameters()))
        out.append((fired, same))
    # the same with gradients kept across steps (zero_grad(set_to_none=False)): the late gradient is then ADDED IN PLACE to
    # the tensor whose copy already left -- nothing to separate any more: every rank raises (the flag rides the host
    # exchange), none hangs in a collective
    net.zero_grad(set_to_none=False)
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(900 + rank))
    fwd(net, x, "late" if rank == 1 else "both", True).backward()
    try:
        ex.finish()
        raised = False
    except RuntimeError as e:
        raised = "early bucket had left" in str(e)
    q.put((rank, out, raised))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_keeps_gradients_produced_after_the_early_bucket_left():
    """a gradient of the early module that appears after the hook fired must not be dropped (the rank
    would keep its local un-reduced value and the ranks diverge silently): produced late -> one more symmetric
    all-reduce, sums equal average_gradients; accumulated in place late -> RuntimeError on EVERY rank."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_late_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, raised in res:
        assert all(same for _, same in out), (rank, out)
        assert [f for f, _ in out] == [False, True, True], (rank, out)
        assert raised, rank


def test_syncbn_exchange_group_is_unchanged_off_nccl():
    """exchange_group(): explicit groups and non-NCCL backends pass through (the gloo tests above run on the default group)."""
    import socket
    import torch.distributed as dist
    from apex import parallel
    if not hasattr(parallel, "exchange_group"):
        pytest.skip("real apex installed")
    sentinel = object()
    assert parallel.exchange_group(sentinel) is sentinel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
    try:
        parallel._HP_GROUP.update(tried=False, group=None)
        assert parallel.exchange_group(None) is None
    finally:
        parallel._HP_GROUP.update(tried=False, group=None)
        dist.destroy_process_group()


@pytest.mark.gpu
def test_syncbn_exchange_group_on_rccl_single_rank():
    """With RSLO_SYNCBN_HP_GROUP=1 on the NCCL / RCCL backend the SyncBN exchanges get their own high-priority group and an
    all-reduce through it works; run in a child process (its own process-group state).  Multi-rank behaviour is the
    backend's: no second GPU on the test box."""
    import subprocess, sys
    code = (
        "import os, sys, socket; sys.path.insert(0, %r); import torch, rslo_amd; import torch.distributed as dist\n"
        "from apex import parallel\n"
        "s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, init_method='tcp://127.0.0.1:%%d' %% port)\n"
        "g = parallel.exchange_group(None)\n"
        "assert g is not None and g is not dist.group.WORLD and parallel.exchange_group(None) is g\n"
        "t = torch.arange(8, dtype=torch.float64, device='cuda'); dist.all_reduce(t, group=g); torch.cuda.synchronize()\n"
        "assert t.tolist() == list(range(8)); dist.destroy_process_group(); print('ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RSLO_SYNCBN_HP_GROUP="1"))
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def _apex_ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    from apex.parallel import DistributedDataParallel as ApexDDP
    from rslo.utils.distributed_utils import average_gradients
    torch.manual_seed(100 + rank)                 # different initial weights per rank: the wrapper broadcasts rank 0's
    net = torch.nn.Module()
    net.encoder = torch.nn.Linear(6, 8)
    net.odom_predictor = torch.nn.Linear(8, 3)    # the attribute the wrapper takes as the early bucket
    net.unused = torch.nn.Linear(3, 3)            # never receives a gradient (77 of the real network's 290 tensors)
    net.forward = lambda x: {"loss": net.odom_predictor(net.__dict__["_grad_exchange"].watch(net.encoder(x))).square().sum(),
                             "aux": torch.zeros(1)}
    wrapped = ApexDDP(net)
    w0 = [p.detach().clone() for p in net.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, [w.tolist() for w in w0])
    same_init = gathered[0] == gathered[1]
    ref = copy.deepcopy(net)
    ref.__dict__.pop("_grad_exchange", None)
    out = []
    for step in range(3):
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * step + rank))
        net.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
        wrapped(x)["loss"].mean().backward()               # train_hdf5.py:623-665: reduced gradients when backward returns
        ref.odom_predictor(ref.encoder(x)).square().sum().backward()
        average_gradients(ref, mean=True)
        out.append(all((a.grad is None and b.grad is None) or torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-7)
                       for a, b in zip(net.parameters(), ref.parameters())))
    q.put((rank, same_init, out, net.unused.weight.grad is None))
    dist.barrier()
    dist.destroy_process_group()


def test_apex_ddp_stand_in_broadcasts_and_averages_inside_backward():
    """compat apex.parallel.DistributedDataParallel (what train_hdf5.py:463 constructs): rank 0's weights everywhere after
    construction; after `loss.backward()` every rank holds the rank-averaged gradients (== average_gradients(mean=True)),
    with the head's bucket overlapped from step 1 on; parameters without a gradient stay without one."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_apex_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_init, out, unused_none in res:
        assert same_init and all(out) and unused_none, (rank, same_init, out, unused_none)
