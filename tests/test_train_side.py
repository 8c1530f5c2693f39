"""Training-side row: optimizer builder + OptimWrapper + OneCycle + checkpoint index against golden
vectors produced by the reference's own modules (tests/golden/make_golden_train.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import rslo_amd  # noqa: F401

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from make_golden_train import TOTAL_STEP, run, scripted_grads, tiny_net   # noqa: E402  (the scripted toy problem, not reference code)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "train_side.npz")), json.load(open(os.path.join(GOLD, "train_side.json")))


def shipped_optimizer_cfg():
    from rslo.utils import config_text
    return config_text.shipped_config().train_config.optimizer


def test_one_cycle_adam_trajectory_matches_reference(gold):
    from rslo.builder import lr_scheduler_builder, optimizer_builder
    g, meta = gold
    assert meta["total_step"] == TOTAL_STEP
    net = tiny_net()
    opt, lrs, moms, traj = run(optimizer_builder, lr_scheduler_builder, shipped_optimizer_cfg(), net)
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(moms, g["moms"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(traj, g["traj"], rtol=2e-6, atol=1e-7)
    sd = opt.state_dict()
    lay = meta["layout"]
    assert [len(x["params"]) for x in sd["param_groups"]] == lay["group_sizes"]      # 4 layer groups x (non-BN, BN)
    assert sorted(sd["state"][0].keys()) == lay["state_keys"]
    assert [x["weight_decay"] for x in sd["param_groups"]] == lay["weight_decay_in_groups"]
    assert opt.name == lay["name"] == "adam_optimizer"
    assert set(lay["group_keys"]) <= set(sd["param_groups"][0].keys()) | {"params"}


def test_shipped_network_gets_eight_param_groups():
    from rslo.builder import optimizer_builder
    from test_golden_host import build_net
    net = build_net()
    opt = optimizer_builder.build(shipped_optimizer_cfg(), net)
    sizes = [len(g["params"]) for g in opt.param_groups]
    assert len(sizes) == 8 and sizes[0] == sizes[1] == 0                 # the VFE has no parameters
    assert sum(sizes) == sum(1 for p in net.parameters() if p.requires_grad) == 288
    assert sizes[7] == 0 and sizes[6] == 2                               # losses: 2 learnable alphas, no BN
    # nn.BatchNorm1d of the covariance branch is the only "BN" half of the middle group (SyncBN is not in bn_types)
    assert sizes[3] == 10 and sizes[5] == 0


def test_checkpoint_index_matches_reference(tmp_path, gold):
    import torchplus.train as T
    _, meta = gold
    net = tiny_net()
    net.name = "voxelnet"
    from rslo.builder import optimizer_builder
    opt = optimizer_builder.build(shipped_optimizer_cfg(), net)
    for step in (10, 20, 30, 40, 50):
        T.save_models(str(tmp_path), [net, opt], step, max_to_keep=3)
    assert json.load(open(tmp_path / "checkpoints.json")) == meta["ckpt_index"]
    assert sorted(os.listdir(tmp_path)) == meta["ckpt_files"]
    assert os.path.basename(T.latest_checkpoint(str(tmp_path), "voxelnet")) == meta["ckpt_latest"]
    # restore round trip, by name
    net2 = tiny_net()
    net2.name = "voxelnet"
    with torch.no_grad():
        for p in net2.parameters():
            p.add_(1.0)
    T.try_restore_latest_checkpoints(str(tmp_path), [net2])
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        T.restore_latest_checkpoints(str(tmp_path), {"missing": net2})
    T.restore_models(str(tmp_path), [net2], 40)


def test_host_copy_checkpoints_keep_state_dict_metadata(tmp_path):
    """save_models_cpu serialises host copies; like the reference's model.cpu().state_dict() (checkpoint.py:178-218)
    they carry the OrderedDict's `_metadata` (per-module versions), so loading takes the same path as for its files."""
    import torchplus.train as T
    net = tiny_net()
    net.name = "voxelnet"
    T.save_models_cpu(str(tmp_path), [net], 3)
    sd = torch.load(T.latest_checkpoint(str(tmp_path), "voxelnet"), weights_only=False)
    assert getattr(sd, "_metadata", None) == net.state_dict()._metadata
    assert list(sd) == list(net.state_dict())


def test_other_schedules_known_answers():
    from torchplus.train import learning_schedules_fastai as lsf

    class Fake:
        lr = 0
        mom = 0
    o = Fake()
    s = lsf.ManualStepping(o, 100, [0.8, 0.9], [1e-3, 1e-4, 5e-5])
    got = []
    for i in (0, 79, 80, 89, 90, 99):
        s.step(i)
        got.append(o.lr)
    assert got == [1e-3, 1e-3, 1e-4, 1e-4, 5e-5, 5e-5]
    s = lsf.ExponentialDecay(o, 100, 3e-4, 0.1, 0.8, staircase=True)
    s.step(0); a = o.lr
    s.step(10); b = o.lr
    s.step(25); c = o.lr
    assert a == 3e-4 and abs(b - 3e-4 * 0.8) < 1e-18 and abs(c - 3e-4 * 0.64) < 1e-18
    s = lsf.LRSchedulerStep(o, 10, [(0, "lambda p: 1.0 + p"), (0.5, "lambda p: 3.0")], [(0, "lambda p: 0.9")])
    s.step(2); assert abs(o.lr - 1.4) < 1e-12 and o.mom == 0.9
    s.step(7); assert o.lr == 3.0


@pytest.mark.gpu
def test_gpu_optimizer_trajectory_matches_reference_vectors(gold, hip):
    """f1 on the device: the same scripted toy problem as the reference-generated trajectory, run on cuda:0 through the
    builders (fused multi-tensor Adam behind OptimWrapper + OneCycle).  Learning rates / momenta exact, parameters
    2e-6 relative of the reference's CPU trajectory."""
    from rslo.builder import lr_scheduler_builder, optimizer_builder
    g, _ = gold
    net = tiny_net().cuda()
    opt, lrs, moms, traj = run(optimizer_builder, lr_scheduler_builder, shipped_optimizer_cfg(), net)
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(moms, g["moms"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(traj, g["traj"], rtol=2e-5, atol=1e-6)
    assert all(p.is_cuda for grp in opt.param_groups for p in grp["params"])


@pytest.mark.gpu
def test_cpu_checkpoints_of_a_gpu_run_hold_host_tensors(tmp_path, hip):
    """save_models_cpu (reference: checkpoint.py:178-218): .tckpt files of a GPU run load without map_location."""
    import torchplus.train as T
    from rslo.builder import optimizer_builder
    net = tiny_net().cuda()
    net.name = "voxelnet"
    opt = optimizer_builder.build(shipped_optimizer_cfg(), net)
    scripted_grads(net, 0)
    opt.step()
    T.save_models_cpu(str(tmp_path), [net, opt], 1)

    def tensors(o):
        if isinstance(o, torch.Tensor):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from tensors(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                yield from tensors(v)
    for f in os.listdir(tmp_path):
        if f.endswith(".tckpt"):
            sd = torch.load(str(tmp_path / f), weights_only=False)
            ts = list(tensors(sd))
            assert ts and all(not t.is_cuda for t in ts), f
    assert next(net.parameters()).is_cuda                      # the live model stayed on the device


@pytest.mark.gpu
def test_hip_clip_and_adam_match_torch(hip):
    """csrc/optim.hip against torch's own formulation on the same tensors: clip_grad_norm_ + decoupled decay + fused
    Adam, 4 param groups with different hyper-parameters, ragged tensor sizes, tensors without gradients, 6 steps
    (the first ones clip, the later ones do not), then a state_dict round trip."""
    from rslo_amd import optim as hip_optim
    torch.manual_seed(3)
    shapes = [(7,), (64, 64, 3, 3), (4097,), (5, 13), (1,), (8192,), (33, 3), (16,)]
    groups_of = [0, 0, 1, 1, 2, 2, 3, 3]
    hyp = [dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8), dict(lr=1e-3, betas=(0.85, 0.95), eps=1e-8),
           dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-6), dict(lr=2e-3, betas=(0.8, 0.9), eps=1e-8)]
    wds = [0.01, 0.0, 0.001, 0.05]

    def make():
        ps = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
        return ps, torch.optim.Adam([dict(params=[p for p, g in zip(ps, groups_of) if g == gi], **hyp[gi])
                                     for gi in range(4)], fused=True)
    torch.manual_seed(5)
    pa, oa = make()
    torch.manual_seed(5)
    pb, ob = make()
    st = hip_optim.stepper_of(oa)
    assert st is not None
    no_grad = {4}                       # a tensor that never gets a gradient
    for it in range(6):
        scale = 40.0 if it < 3 else 0.01
        for i, (a, b) in enumerate(zip(pa, pb)):
            a.grad = b.grad = None
            # tensor 1 gets its FIRST gradient at step 3, tensor 5 skips step 2: torch.optim.Adam counts per tensor
            if i in no_grad or (i == 1 and it < 3) or (i == 5 and it == 2):
                continue
            g = torch.randn_like(a) * scale
            a.grad, b.grad = g.clone(), g.clone()
        ta = hip_optim.clip_grad_norm_(pa, 10.0, optimizer=oa)
        tb = torch.nn.utils.clip_grad_norm_(pb, 10.0)
        assert abs(float(ta) - float(tb)) <= 2e-6 * float(tb)
        for a, b in zip(pa, pb):
            if a.grad is not None:
                assert torch.allclose(a.grad, b.grad, rtol=2e-6, atol=0)
        assert st.step(wds)
        with torch.no_grad():
            for gi, grp in enumerate(ob.param_groups):
                torch._foreach_mul_(list(grp["params"]), 1 - wds[gi] * grp["lr"])
        ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), it
    assert [float(oa.state[a]["step"]) for a in pa if oa.state.get(a)] == [6.0, 3.0, 6.0, 6.0, 5.0, 6.0, 6.0]
    for a, b in zip(pa, pb):
        if a.grad is None:
            assert a not in oa.state        # no empty entry for a never-stepped tensor (checkpoint parity)
            continue
        assert float(oa.state[a]["step"]) == float(ob.state[b]["step"])
        assert torch.allclose(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"], rtol=1e-5, atol=1e-8)
        assert torch.allclose(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-10)
    # a state_dict round trip replaces the moment tensors: the tables follow
    oa.load_state_dict(oa.state_dict())
    for a, b in zip(pa, pb):
        if a.grad is not None:
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
    assert st.step(wds)
    with torch.no_grad():
        for gi, grp in enumerate(ob.param_groups):
            torch._foreach_mul_(list(grp["params"]), 1 - wds[gi] * grp["lr"])
    ob.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
        if a.grad is not None:
            assert float(oa.state[a]["step"]) == float(ob.state[b]["step"])      # 7; 4 (late tensor); 6 (skipped once)


def test_head_graph_host_logic_without_a_gpu():
    """rslo_amd/headgraph.py: the prediction dict survives flatten / unflatten (nested lists, Nones, constants); a CPU call
    is never routed to the graphs (the eager pass is the path there); streams.join_in_enclosing_pass nests and queues nothing
    when no leaf work is pending."""
    import torch
    from rslo_amd import headgraph, streams
    a, b, c = torch.zeros(2), torch.ones(3), torch.full((1,), 2.0)
    tree = {"t": [a], "pyr": [[b, c], [c, None]], "k": 3, "tuple": (a, "x")}
    flat = []
    spec = headgraph._flatten(tree, flat)
    assert len(flat) == 5 and all(torch.is_tensor(t) for t in flat)
    back = headgraph._unflatten(spec, [t + 1 for t in flat])
    assert back["k"] == 3 and back["pyr"][1][1] is None and back["tuple"][1] == "x"
    assert torch.equal(back["pyr"][0][1], c + 1) and torch.equal(back["t"][0], a + 1)

    class _Head(torch.nn.Module):
        graph_capturable = True
    h = _Head().train()
    x = torch.zeros(1, 4, 2, 2, requires_grad=True)
    assert not headgraph.wanted(h, x, 2)            # never for a CPU tensor
    old = headgraph.ENABLED
    headgraph.ENABLED = True
    try:
        assert not headgraph.wanted(h, x, 2) and not headgraph.wanted(h, None, 2)
    finally:
        headgraph.ENABLED = old
    with streams.join_in_enclosing_pass():
        with streams.join_in_enclosing_pass():
            assert streams._hold[0] == 2
    assert streams._hold[0] == 0
