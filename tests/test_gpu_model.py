"""GPU parity of the assembled path: host mirror + HIP kernels vs (a) golden vectors from the real reference
and (b) the same modules run on the CPU over the oracle backend (oracle/cpu_backend.py).

Tolerances: poses 1e-4 relative (the north star's bar), losses 5e-4 relative, features/gradients fp32-level.
"""
import copy
import os

import numpy as np
import pytest
import torch

import rslo_amd  # noqa: F401
from oracle import cpu_backend
from rslo_amd import synthetic, workload

from test_golden_host import (check_closs, check_create_loss, check_head, check_head_variant, check_normalisation_variant,
                              check_sparse_conv_2d, check_vote_svd, load_small_head, make_closs)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pieces():
    return np.load(os.path.join(GOLD, "ref_pieces.npz"))


def test_consistency_loss_gpu_matches_reference(hip, pieces):
    closs = make_closs().cuda()          # real HIP chamfer inside
    check_closs(pieces, closs, "cuda")


def test_fused_loss_kernels_match_torch_formulation(hip, pieces):
    """rslo_cov_residual_fwd/bwd + rslo_icp_step vs the op-by-op torch formulation on the same GPU inputs."""
    from test_golden_host import run_closs
    res = {}
    for fused in (True, False):
        closs = make_closs().cuda()
        closs.use_fused = fused
        l, rr, tt, leaves = run_closs(pieces, closs, "cuda", 2)
        l.backward()
        res[fused] = (l.detach(), rr, tt, [t.grad.clone() for t in leaves])
    a, b = res[True], res[False]
    assert float((a[0] - b[0]).abs() / b[0].abs()) < 1e-5
    assert float((a[1] - b[1]).abs().max()) < 1e-5 and float((a[2] - b[2]).abs().max()) < 2e-5
    for ga, gb, name in zip(a[3], b[3], ("cov1", "cov2", "q", "T")):
        assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max()) + 1e-8, name


def test_icp_step_recovers_known_motion(hip):
    """SVDHead known answer: a cloud and its rigidly moved copy -> the inverse motion."""
    import kornia
    g = torch.Generator(device="cuda").manual_seed(0)
    N = 5000
    p = torch.randn(2, N, 3, device="cuda", generator=g) * torch.tensor([20.0, 10.0, 2.0], device="cuda")
    n = torch.nn.functional.normalize(torch.randn(2, N, 3, device="cuda", generator=g), dim=-1)
    R = kornia.quaternion_to_rotation_matrix(torch.tensor([[0.01, -0.02, 0.03, 1.0], [0.0, 0.0, -0.05, 1.0]], device="cuda"))
    t = torch.tensor([[0.7, 0.1, -0.05], [-0.4, 0.3, 0.02]], device="cuda")
    tgt = (p @ R.transpose(1, 2) + t[:, None]).contiguous()
    idx = torch.arange(N, dtype=torch.int32, device="cuda").repeat(2, 1).contiguous()
    dist = torch.zeros(2, N, device="cuda")
    thr = torch.ones(2, device="cuda")
    res_r = torch.eye(3, device="cuda").repeat(2, 1, 1)
    res_t = torch.zeros(2, 3, device="cuda")
    hip.icp_step(p.contiguous(), n.contiguous(), tgt, idx, dist, thr, res_r, res_t)
    # returns the map target -> source: R^T, -R^T t
    assert float((res_r - R.transpose(1, 2)).abs().max()) < 1e-5
    assert float((res_t + (R.transpose(1, 2) @ t[..., None]).squeeze(-1)).abs().max()) < 1e-4
    moved = hip.transform_points(tgt, res_r, res_t)
    assert float((moved - p).abs().max()) < 1e-3
    # first=True: the kernel starts from the identity itself -- same bits as the filled start, whatever the buffers held
    fr = torch.full((2, 3, 3), float("nan"), device="cuda")
    ft = torch.full((2, 3), float("nan"), device="cuda")
    hip.icp_step(p.contiguous(), n.contiguous(), tgt, idx, dist, thr, fr, ft, first=True)
    assert torch.equal(fr, res_r) and torch.equal(ft, res_t)


def test_bev_head_gpu_matches_reference(hip):
    head, g = load_small_head()
    check_head(head, g, "cuda")


def test_create_loss_gpu_matches_reference(hip, pieces):
    check_create_loss(pieces, "cuda", None)


def test_fused_pyramid_loss_matches_torch_formulation(hip):
    """rslo_pyramid_l2_fwd/_bwd vs the reference formulation op by op (gen_tq_maps -> nearest interpolate ->
    AdaptiveWeightedL2Loss per level): values 1e-5 rel, gradients 1e-4 rel; includes a non-integer resampling ratio."""
    torch.manual_seed(11)
    net, _ = workload.build_network()
    net.train()
    B = 3
    sizes = [(24, 44), (40, 88), (96, 176)]
    tq = torch.randn(B, 7, device="cuda")
    tq[:, 3:] = torch.nn.functional.normalize(tq[:, 3:] + torch.tensor([3.0, 0, 0, 0], device="cuda"), dim=-1)
    with torch.no_grad():
        net._pyramid_translation_loss.alpha.fill_(0.3)
        net._pyramid_rotation_loss.alpha.fill_(-0.2)
    preds = [torch.randn(B, 7, h, w, device="cuda", requires_grad=True) for h, w in sizes]
    masks = [torch.rand(B, 2, h, w, device="cuda") * (torch.rand(B, 1, h, w, device="cuda") > 0.4) for h, w in sizes]
    outs = []
    for fused in (True, False):
        net.fused_pyramid = fused
        example = {"icp_odometry": tq.clone(), "tq_maps": None, "num_voxels": [torch.zeros(B, 1)] * 2}
        pd = {"translation_preds": [tq[:, :3] * 1.1], "rotation_preds": [tq[:, 3:]],
              "pyramid_motion": [[p, m] for p, m in zip(preds, masks)]}
        cl, net._consistency_loss = net._consistency_loss, None
        try:
            ret = net.loss(example, pd)
        finally:
            net._consistency_loss = cl
        grads = torch.autograd.grad(ret["loss"].sum(), preds +
                                    [net._pyramid_translation_loss.alpha, net._pyramid_rotation_loss.alpha])
        outs.append((ret["pyramid_loss"], grads))
    assert rel(outs[0][0], outs[1][0]) < 1e-5
    for a, b in zip(outs[0][1], outs[1][1]):
        assert rel(a, b) < 1e-4


def reduced_pair(seed=0, rings=16):
    """A cheap KITTI-shaped pair: every (64/rings)-th ring of the synthetic scan (~8k voxels/frame)."""
    p0, p1, motion = synthetic.frame_pair(seed)
    rings_id0 = np.arange(len(p0)) * 64 // len(p0)
    rings_id1 = np.arange(len(p1)) * 64 // len(p1)
    step = 64 // rings
    return p0[rings_id0 % step == 0], p1[rings_id1 % step == 0], motion


def clone_to_cpu(net):
    net_cpu = copy.deepcopy(net).cpu()
    return net_cpu


def example_to_cpu(ex):
    out = {}
    for k, v in ex.items():
        if isinstance(v, list):
            out[k] = [x.cpu() if isinstance(x, torch.Tensor) else x for x in v]
        elif isinstance(v, torch.Tensor):
            out[k] = v.cpu()
        else:
            out[k] = v
    return out


def example_to_f64(ex):
    f = lambda x: x.double() if isinstance(x, torch.Tensor) and x.is_floating_point() else x   # noqa: E731
    return {k: ([f(x) for x in v] if isinstance(v, list) else f(v)) for k, v in ex.items()}


def bias_before_bn(net):
    """Names of conv biases that feed straight into a BatchNorm: their gradient is analytically zero (BN removes the
    mean), so both sides hold pure rounding noise and cannot be compared."""
    names = set()
    for mname, m in net.named_modules():
        kids = list(m.named_children())
        for (n0, c0), (n1, c1) in zip(kids, kids[1:]):
            if isinstance(c1, torch.nn.modules.batchnorm._BatchNorm) and getattr(c0, "bias", None) is not None \
                    and not isinstance(c0, torch.nn.modules.batchnorm._BatchNorm):
                names.add((mname + "." if mname else "") + n0 + ".bias")
    return names


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def trained_like_init(net):
    """Deterministic stand-in for a trained head.  A random-init head votes a pose that is metres / radians off and
    spreads its softmax confidences by chance; the ICP rounds of the loss are chaotic from there and the vote amplifies
    fp32 rounding of the logits -- comparisons between ANY two implementations are only meaningful inside ICP's basin.
    Here the per-unit (t, q) regressor starts at a plausible ego-motion (0.8 m forward, identity rotation) with a small
    data-dependent part (the un-normalised trunk features are O(100): 1e-3 keeps it at centimetres), and both
    confidence heads start near uniform."""
    with torch.no_grad():
        last = net.odom_predictor.tq_map_conv[6]
        last.weight.mul_(1e-3)
        last.bias.copy_(torch.tensor([0.8, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]))
        for conf in (net.odom_predictor.t_map_conf, net.odom_predictor.q_map_conf):
            conf.conf_model[6].weight.mul_(1e-2)


def test_encoder_fwd_bwd_matches_cpu_oracle(hip):
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    p0, p1, _ = reduced_pair(0)
    ex = workload.make_example(net, [[p0, p1]])
    feats = [net.voxel_feature_extractor(ex["voxels"][t], ex["num_points"][t]) for t in range(2)]
    coords = torch.cat([ex["coordinates"][0], ex["coordinates"][1] + torch.tensor([1, 0, 0, 0], dtype=torch.int32, device="cuda")], 0)
    x = torch.cat(feats, 0)
    enc = net.middle_feature_extractor
    bev, cov = enc(x, coords, 2)
    g_bev, g_cov = torch.randn_like(bev), torch.randn_like(cov)
    (bev * g_bev).sum().add((cov * g_cov).sum()).backward()

    enc_cpu = copy.deepcopy(enc).cpu()
    enc_cpu.zero_grad()
    with cpu_backend.patched():
        bev_c, cov_c = enc_cpu(x.cpu(), coords.cpu(), 2)
        ((bev_c * g_bev.cpu()).sum() + (cov_c * g_cov.cpu()).sum()).backward()
    assert bev.shape == (2, 128, 96, 176)
    assert rel(bev, bev_c) < 2e-5 and rel(cov, cov_c) < 2e-5
    # weight gradients: 20 chained fp32 layers vs the oracle's double accumulation, plus LeakyReLU masks that can
    # flip for |y| ~ 1e-7 -- a few 1e-3 of the largest entry on the earliest layers
    skip = bias_before_bn(enc)
    assert len(skip) == 5
    for (n, p), (_, pc) in zip(enc.named_parameters(), enc_cpu.named_parameters()):
        if n not in skip:
            assert rel(p.grad, pc.grad) < 5e-3, n


def test_c5_dense_scan_encoder_fwd_bwd_matches_cpu_oracle(hip):
    """BASELINE config C5, per-GPU part (the 20-layer GU encoder + covariance branch of
    rslo/models/middle.py:119-245 on 128-ring scans -- 263 k points -- at 0.1 m cubic voxels, sparse shape
    [81, 768, 1408], bs 2 = the per-GPU batch of the config): voxelizer + native plan + forward + backward against the
    same modules over the CPU oracle.  BEV / covariance head 2e-5, weight gradients 5e-3 of the largest entry."""
    import spconv
    from rslo.models import middle
    from rslo_amd import plan as native_plan
    torch.manual_seed(11)
    gen = spconv.utils.VoxelGenerator(list(synthetic.VOXEL_SIZE_DENSE), list(synthetic.PC_RANGE),
                                      synthetic.MAX_POINTS_PER_VOXEL, 1 << 18)
    enc = middle.get_middle_class("SpMiddleFHDWithCov2_3")(
        [1] + gen.grid_size[::-1].tolist() + [7], bn_type="None", use_leakyReLU=True, num_input_features=7,
        num_filters_down1=[], num_filters_down2=[]).cuda().train()
    assert list(enc.sparse_shape) == [81, 768, 1408]
    clouds = [torch.from_numpy(synthetic.scan(n_el=128, scan_seed=i)).cuda() for i in range(2)]
    res = gen.generate_many(clouds, 1 << 18)
    feats = torch.cat([hip.vfe_mean(v, n) for v, c, n in res], 0)
    coords = torch.cat([torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device="cuda"), c], 1)
                        for b, (v, c, n) in enumerate(res)], 0)
    assert clouds[0].shape[0] > 260000 and feats.shape[0] > 100000
    x = feats.clone().requires_grad_(True)
    bev, cov = enc(x, coords, 2)
    assert bev.shape == (2, 64 * 4, 96, 176)            # C * D = 256 channels at this voxel size
    g_bev, g_cov = torch.randn_like(bev), torch.randn_like(cov)
    ((bev * g_bev).sum() + (cov * g_cov).sum()).backward()
    # the site sets of all five levels, as the oracle's rulebook chain gives them
    plan = enc.plan(coords, 2)
    n_lvl = [plan.indices.shape[0]] + [plan.indice_dict[k].out_index.coords.shape[0]
                                        for k in ("conv3d2", "conv3d3", "conv3d4", "conv3d5")]
    enc_cpu = copy.deepcopy(enc).cpu()
    enc_cpu.zero_grad()
    xc = feats.cpu().clone().requires_grad_(True)
    with cpu_backend.patched():
        bev_c, cov_c = enc_cpu(xc, coords.cpu(), 2)
        ((bev_c * g_bev.cpu()).sum() + (cov_c * g_cov.cpu()).sum()).backward()
        plan_c = enc_cpu.plan(coords.cpu(), 2)
    assert n_lvl == [plan_c.indices.shape[0]] + [plan_c.indice_dict[k].out_index.coords.shape[0]
                                                 for k in ("conv3d2", "conv3d3", "conv3d4", "conv3d5")]
    assert n_lvl[0] > n_lvl[1] > n_lvl[2] > n_lvl[3] > n_lvl[4] > 3000
    assert rel(bev, bev_c) < 2e-5 and rel(cov, cov_c) < 2e-5
    assert rel(x.grad, xc.grad) < 2e-3
    # weight gradients against the float64 arbiter (the same modules in double): 5e-3 of the largest entry; the CPU
    # fp32 path's own distance to the arbiter is measured beside it -- the covariance branch's first layer feeds a
    # BatchNorm over 56 k rows and a LeakyReLU whose mask flips for |y| ~ 1e-7, both fp32 paths sit at ~3e-3 there
    enc_f64 = copy.deepcopy(enc).cpu().double()
    enc_f64.zero_grad()
    with cpu_backend.patched():
        bev_d, cov_d = enc_f64(feats.cpu().double(), coords.cpu(), 2)
        ((bev_d * g_bev.cpu().double()).sum() + (cov_d * g_cov.cpu().double()).sum()).backward()
    assert rel(bev, bev_d) < 2e-5 and rel(cov, cov_d) < 2e-5
    skip = bias_before_bn(enc)
    rows = [(rel(p.grad, pd.grad), rel(pc.grad, pd.grad), n) for (n, p), (_, pc), (_, pd) in
            zip(enc.named_parameters(), enc_cpu.named_parameters(), enc_f64.named_parameters()) if n not in skip]
    print("C5 weight gradients vs float64: GPU max %.2e (%s), CPU fp32 max %.2e, medians %.2e / %.2e" % (
        max(rows)[0], max(rows)[2], max(r[1] for r in rows), np.median([r[0] for r in rows]),
        np.median([r[1] for r in rows])))
    # measured: GPU max 1.96e-2 (middle_cov_deconv.9.weight; the CPU fp32 path: 1.97e-2 on the same tensor), medians
    # 1.7e-3 (GPU) / 2.5e-3 (CPU fp32): where a tensor is past 5e-3 BOTH fp32 paths are (BatchNorm statistics over
    # 56 k rows + LeakyReLU masks of the covariance branch), so the bar there is the CPU path's own distance x 1.5
    for e_gpu, e_cpu, n in rows:
        assert e_gpu < max(5e-3, 1.5 * e_cpu), (n, e_gpu, e_cpu)
    assert np.median([r[0] for r in rows]) < 1.5 * np.median([r[1] for r in rows])


def three_way(net, ex):
    """One training step three ways on the same inputs and parameters: HIP kernels (fp32), the same host modules over
    the CPU oracle in fp32 (the reference-semantics path), and over the oracle backend in float64 (the arbiter that
    separates rounding from errors).  Returns [(outputs, network)] in that order."""
    net_c32 = copy.deepcopy(net).cpu()
    net_f64 = copy.deepcopy(net).cpu().double()
    ret = net(ex)
    ret["loss"].backward()
    ex_cpu = example_to_cpu(ex)
    with cpu_backend.patched():
        ret_c = net_c32(ex_cpu)
        ret_c["loss"].backward()
        ret_64 = net_f64(example_to_f64(ex_cpu))
        ret_64["loss"].backward()
    return (ret, net), (ret_c, net_c32), (ret_64, net_f64)


def gradient_errors(nets, skip):
    """Per parameter: (max-rel error of the GPU gradient vs float64, of the CPU-fp32 gradient vs float64, name)."""
    (gpu, c32, f64) = nets
    rows = []
    for (n, p), (_, pc), (_, p64) in zip(gpu.named_parameters(), c32.named_parameters(), f64.named_parameters()):
        if n in skip:
            continue
        if p64.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, n
        if float(p64.grad.abs().max()) < 1e-6:    # analytically zero (e.g. the softmax-shift bias of a confidence head)
            assert float(p.grad.abs().max()) < 1e-5, n
            continue
        l2 = float((p.grad.detach().cpu().double() - p64.grad).norm() / (p64.grad.norm() + 1e-30))
        rows.append((rel(p.grad, p64.grad), rel(pc.grad, p64.grad), n, l2))
    return rows


def check_three_way(net, ex, median_bar, max_bar, ratio_bar, pose_bar=1e-5):
    """Bars (measured values in DESIGN.md section 4; scripts/parity_report.py prints the full table):
      * poses within 1e-5 relative of the CPU path AND of the float64 arbiter (measured 2e-7; north star: 1e-4),
      * loss terms within 1e-4 relative (measured <= 1.2e-5: C_loss, which sees the pose through residuals of ~0.1 m
        on coordinates of ~50 m),
      * gradients against float64.  The consistency loss differentiates residuals of ~0.1 m between points ~50 m from
        the sensor, so a pose that differs by 1e-6 (fp32 rounding of either implementation) moves every gradient
        by ~1e-3: BOTH fp32 paths sit at 3e-4..3e-3 from float64, which is why the bar is on the distance to the
        arbiter and on the ratio to the CPU path's own distance, not on GPU-vs-CPU alone."""
    (ret, _), (ret_c, _), (ret_64, _) = res = three_way(net, ex)
    for k in ("translation_preds", "rotation_preds"):
        assert rel(ret[k], ret_c[k]) < pose_bar, k
        assert rel(ret[k], ret_64[k]) < pose_bar, k
    for k in ("loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
        assert rel(ret[k], ret_c[k]) < 1e-4, k
        assert rel(ret[k], ret_64[k]) < 1e-4, k
    rows = gradient_errors([r[1] for r in res], bias_before_bn(net))
    assert len(rows) >= 170   # 213 of 290 parameter tensors receive gradients
    e_gpu = np.array([r[0] for r in rows])
    e_cpu = np.array([r[1] for r in rows])
    assert np.median(e_gpu) < median_bar, (np.median(e_gpu), np.median(e_cpu))
    assert np.median(e_gpu) < ratio_bar * np.median(e_cpu), (np.median(e_gpu), np.median(e_cpu))
    # per tensor: within max_bar of float64 (largest error over largest entry) -- or an EXCEPTION that the table below
    # justifies in one of two ways: (i) the CPU fp32 path (reference-semantics arithmetic, no HIP kernel involved) is itself at
    # least half as far from float64 on the very same tensor -- ill-conditioned in fp32 for ANY implementation; (ii) the error
    # is CONCENTRATED: the tensor's relative L2 error stays under 1e-2 while single entries stand out -- the signature of a
    # ReLU / LeakyReLU mask that flipped for an activation within rounding of zero (one pixel's outer product lands in the
    # weight gradient), which a mis-computed kernel (wrong tap, wrong channel block, lost partial sum) cannot produce: that
    # moves the L2 error with the maximum.  Nothing may be past 0.1 under any excuse.
    over = sorted((r for r in rows if r[0] >= max_bar), reverse=True)
    if over:
        print("tensors past %.0e of float64 (GPU max-err, CPU-fp32 max-err, GPU rel-L2 err, name):" % max_bar)
        for e_g, e_c, n, l2 in over:
            print("  %.2e  %.2e  %.2e  %s" % (e_g, e_c, l2, n))
    print("gradients vs float64: GPU median %.2e max %.2e | CPU fp32 median %.2e max %.2e | %d tensors, %d exceptions" % (
        np.median(e_gpu), e_gpu.max(), np.median(e_cpu), e_cpu.max(), len(rows), len(over)))
    for e_g, e_c, n, l2 in over:
        assert e_g < 0.1 and (e_g <= 2.0 * e_c or l2 <= 1e-2), (n, e_g, e_c, l2)
    assert len(over) <= max(3, len(rows) // 50), over
    assert np.median([r[3] for r in rows]) < median_bar
    return rows


def test_full_network_fwd_bwd_matches_cpu_oracle(hip):
    """Reduced pair (every 4th ring, ~9-10 k voxels per frame), one sample."""
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    trained_like_init(net)
    p0, p1, _ = reduced_pair(1)
    ex = workload.make_example(net, [[p0, p1]])
    # measured: median 2.1e-4 (CPU fp32 path: 2.9e-4), max 3.1e-2 (one BatchNorm bias of the rotation-confidence trunk)
    check_three_way(net, ex, median_bar=1e-3, max_bar=3e-2, ratio_bar=3.0)


def test_c3_full_size_step_matches_cpu_oracle(hip):
    """BASELINE config C3 itself: 4 frame pairs of full 64-ring scans (~130 k voxels in the batched encoder pass),
    forward + loss + backward, against the CPU path and the float64 arbiter (about 90 s of CPU work)."""
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    trained_like_init(net)
    ex = workload.make_example(net, [list(reduced_pair(b + 1, rings=64)[:2]) for b in range(4)])
    assert sum(int(v.sum()) for v in ex["num_voxels"]) > 200000
    # The gradient distance to float64 is set by the consistency loss's sensitivity to 1e-7 pose differences, not by
    # any one kernel: measured over init seeds 7 / 8 / 9 (scripts/parity_report.py --bs 4 --rings 64 --seed N), GPU
    # medians 6.5e-3 / 3.5e-3 / 1.9e-3 with the hand-written stride-2 kernels and 3.7e-3 / 3.6e-3 / 1.7e-3 with the
    # library's (RSLO_CONV2D_S2=0), the CPU fp32 path's own 3.5e-3 / 1.2e-3 / 1.9e-3, median ratio 1.5 / 1.8 / 3.9,
    # max 3.8e-2 (poses agree to 2e-7 in every variant).  Bars (round 4): median 1e-2 and at most 3x the CPU fp32 path's
    # own median; per tensor 3e-2, past that only as a printed exception where the CPU fp32 path is equally far off.
    check_three_way(net, ex, median_bar=1e-2, max_bar=3e-2, ratio_bar=3.0)


def real_training_steps(net, n_steps, clouds_of_step):
    """n_steps of the driver's training step (train_hdf5.py:618-674: OneCycle schedule, backward, clip 10, OptimWrapper /
    Adam, global step) from the network's current state, on the GPU."""
    from rslo.builder import lr_scheduler_builder, optimizer_builder
    from rslo.utils import config_text
    cfg = config_text.shipped_config().train_config
    opt = optimizer_builder.build(cfg.optimizer, net)
    sched = lr_scheduler_builder.build(cfg.optimizer, opt, cfg.steps)
    losses = []
    for i in range(n_steps):
        sched.step(net.get_global_step())
        opt.zero_grad()
        ret = net(workload.make_example(net, clouds_of_step(i)))
        ret["loss"].mean().backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
        opt.step()
        net.update_global_step()
        losses.append(float(ret["loss"].mean()))
    assert all(np.isfinite(losses)), losses
    return losses


POSE_BAR = 3e-5      # a fixed state since round 5 (bit-reproducible training steps): measured 1.7e-5 / 1.6e-5 (translation), 1.2e-5 / 3.4e-6 (rotation)


def test_parity_on_weights_produced_by_real_optimizer_steps(hip):
    """Verdict r3 #4a: the three-way comparison NOT on a hand-made state but on weights a real (short) training run
    produced: seed 7, default init, 50 optimizer steps of the shipped schedule on the GPU from global step 0 (the
    reference's warm-up regime: identity pose inside the consistency loss and icp_iter = 5 while global_step <= 1500,
    voxel_odom_net.py:670-690), batches of two reduced pairs drawn round-robin from six.  Then (1) one more step's
    forward + loss + backward three ways at the state training left (global step 50, running statistics included), and
    (2) the same weights evaluated past the warm-up (global step 2000: predicted pose inside the loss, icp_iter = 2)."""
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    pool = [list(reduced_pair(i)[:2]) for i in range(6)]
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    losses = real_training_steps(net, 50, lambda i: [pool[(2 * i) % 6], pool[(2 * i + 1) % 6]])
    assert net.get_global_step() == 50
    moved = (torch.cat([p.detach().reshape(-1) for p in net.parameters()]) - w0).abs().max()
    assert float(moved) > 1e-3                      # the optimizer really ran
    print("50 real steps: loss %.4f -> %.4f, largest weight change %.2e" % (losses[0], losses[-1], float(moved)))
    ex = workload.make_example(net, [pool[1]])
    net.zero_grad(set_to_none=True)
    # Outputs and loss terms.  The 50 GPU steps are bit-reproducible since round 5 (partner gradients of the covariance residual
    # added in source order), so this is ONE state, the same every run: pose bar 3e-5 (measured 1.7e-5, relative to the largest
    # component of a vote whose components are < 1 m; north star 1e-4), loss terms 1e-4.
    (ret, g), (ret_c, c), (ret_64, d) = three_way(copy.deepcopy(net), ex)
    print("pose vs cpu / f64:", {k: ("%.1e" % rel(ret[k], ret_c[k]), "%.1e" % rel(ret[k], ret_64[k])) for k in ("translation_preds", "rotation_preds")})
    for k in ("translation_preds", "rotation_preds"):
        assert rel(ret[k], ret_c[k]) < POSE_BAR and rel(ret[k], ret_64[k]) < POSE_BAR, k
    for k in ("loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
        assert rel(ret[k], ret_c[k]) < 1e-4 and rel(ret[k], ret_64[k]) < 1e-4, k
    # Gradients.  In this regime the head's gradient is small (regression terms only) and the float64 arbiter ITSELF is
    # hypersensitive: with default-initialised trunk weights and a sparse BEV map, BasicBlock blocks.1.1 holds channels of
    # near-zero variance (invstd ~ 1 / sqrt(eps) = 30) whose activations sit within rounding of the ReLU threshold; moving that
    # block's INPUT by 2.6e-6 (the GPU's forward rounding) flips two of 135 168 mask bits and moves the arbiter's own input
    # gradient by 1.1e-2 in L2 -- measured with the block evaluated in float64 on both inputs (scripts/warmup_parity*.py,
    # profiles/NOTES.md round 4), and everything upstream of that block (stage 0, the encoder tail) inherits it.  Given
    # IDENTICAL inputs the HIP block equals float64 to 1e-6 (test_basic_blocks_are_exact_on_the_inputs_they_see_in_the_network).
    # So the bar here is on what is well-posed: every tensor points the same way as the arbiter's and has its size.
    rows = gradient_errors([g, c, d], bias_before_bn(net))
    assert len(rows) >= 170
    l2 = np.array([r[3] for r in rows])
    print("warm-up regime, trained weights: gradient rel-L2 error vs float64: median %.2e, 90%% %.2e, max %.2e (%s)" % (
        np.median(l2), np.percentile(l2, 90), l2.max(), max(rows, key=lambda r: r[3])[2]))
    assert np.median(l2) < 1.5e-2 and l2.max() < 5e-2, (np.median(l2), l2.max())
    # tensors the flip cannot reach (downstream of blocks.1.1 in backward order: the decoder, the heads, the covariance branch)
    clean = [r for r in rows if any(t in r[2] for t in ("deblocks", "tq_map_conv", "pyramid_motion", "conf_model", "middle_cov_deconv"))]
    assert len(clean) >= 40 and np.median([r[0] for r in clean]) < 2e-4, np.median([r[0] for r in clean])
    net.global_step.fill_(2000)
    net.zero_grad(set_to_none=True)
    (ret, _), (ret_c, _), (ret_64, _) = three_way(copy.deepcopy(net), ex)
    # past the warm-up on these weights the voted pose is still far off (50 steps at lr ~1e-4 do not train a head), so the
    # ICP rounds start outside their basin; with a reproducible state the loss terms are asserted too (round 4 could only print
    # them: measured now 6.8e-5 / 2.9e-6 / 2.1e-5 for translation / rotation / pyramid against float64, C_loss 1.8e-5)
    print("pose past warm-up vs cpu / f64:", {k: ("%.1e" % rel(ret[k], ret_c[k]), "%.1e" % rel(ret[k], ret_64[k])) for k in ("translation_preds", "rotation_preds")})
    for k in ("translation_preds", "rotation_preds"):
        assert rel(ret[k], ret_c[k]) < POSE_BAR and rel(ret[k], ret_64[k]) < POSE_BAR, k
    print("past warm-up on the trained weights, loss terms gpu-f64 / cpu-f64:", {
        k: ("%.1e" % rel(ret[k], ret_64[k]), "%.1e" % rel(ret_c[k], ret_64[k])) for k in ("translation_loss", "rotation_loss",
                                                                                         "pyramid_loss")})
    print("past warm-up on the trained weights: C_loss gpu %.6f cpu %.6f f64 %.6f" % (
        float(ret["C_loss"]), float(ret_c["C_loss"]), float(ret_64["C_loss"])))
    for k in ("translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
        assert rel(ret[k], ret_64[k]) < 1e-4, (k, rel(ret[k], ret_64[k]))


def test_training_steps_are_bit_reproducible(hip):
    """Verdict r4 #4: two runs of the same 30 optimizer steps (seed 7, default init, the shipped schedule, warm-up regime, four
    streams) end in IDENTICAL weights and BatchNorm statistics, bit for bit -- every reduction of the step has a fixed order
    since the partner gradients of the covariance residual are added in source order (rslo_cov_residual_bwd, round 5; with
    the per-run atomics of rounds 1-4 -- deleted in round 6 -- the same two runs differed)."""
    import hashlib
    pool = [list(reduced_pair(i)[:2]) for i in range(6)]

    def run():
        torch.manual_seed(7)
        net, _ = workload.build_network()
        net.train()
        losses = real_training_steps(net, 30, lambda i: [pool[(2 * i) % 6], pool[(2 * i + 1) % 6]])
        torch.cuda.synchronize()
        state = torch.cat([t.detach().double().reshape(-1) for t in list(net.parameters()) +
                           [b for b in net.buffers() if b.is_floating_point()]])
        return hashlib.sha256(state.cpu().numpy().tobytes()).hexdigest(), losses
    h1, l1 = run()
    h2, l2 = run()
    assert l1 == l2, [(a, b) for a, b in zip(l1, l2) if a != b][:3]
    assert h1 == h2
    print("30 steps twice: identical (%s)" % h1[:12])


def test_replayed_head_graphs_train_bit_identically_to_the_eager_pass(hip, monkeypatch):
    """rslo_amd/headgraph.py: 12 optimizer steps with the BEV head's forward / backward replayed from two hipGraphs (captured on
    the third step) end in the SAME weights, BatchNorm statistics (incl. num_batches_tracked) and per-step losses as 12 steps
    issued launch by launch -- same kernels, same order per stream.  Also: gradients that are not reset between two backward
    passes add up (the static gradient buffers are copied out before a replay overwrites them)."""
    import hashlib
    from rslo_amd import headgraph
    pool = [list(reduced_pair(i)[:2]) for i in range(6)]

    def run(graphs):
        monkeypatch.setattr(headgraph, "ENABLED", bool(graphs))
        monkeypatch.setattr(headgraph, "MODE", graphs or None)
        torch.manual_seed(7)
        net, _ = workload.build_network()
        net.train()
        losses = real_training_steps(net, 12, lambda i: [pool[(2 * i) % 6], pool[(2 * i + 1) % 6]])
        torch.cuda.synchronize()
        state = torch.cat([t.detach().double().reshape(-1) for t in list(net.parameters()) + list(net.buffers())])
        st = headgraph._STATE.get(net.odom_predictor)
        return hashlib.sha256(state.cpu().numpy().tobytes()).hexdigest(), losses, st, net
    h0, l0, st0, _ = run(False)
    assert st0 is None
    for mode in ("fwd", "full"):      # forward replayed + backward issued over the retained autograd graph / both replayed
        h1, l1, st1, net = run(mode)
        assert st1 is not None and st1.graph is not None and st1.graph.mode == mode and not st1.failed
        assert l0 == l1, (mode, [(i, a, b) for i, (a, b) in enumerate(zip(l0, l1)) if a != b][:3])
        assert h0 == h1, mode
    # ... and on a non-default stream, where the "fwd" capture sits on the caller's stream and the leaf stream is in use
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        h2, l2, st2, _ = run("fwd")
    torch.cuda.current_stream().wait_stream(s_)
    assert st2.graph.on_callers_stream and l2 == l0 and h2 == h0
    monkeypatch.setattr(headgraph, "MODE", "full")
    # two backward passes without a reset in between: the head's gradients are the sum of the two passes'
    ex = [workload.make_example(net, [pool[0], pool[1]]), workload.make_example(net, [pool[2], pool[3]])]
    w = next(net.odom_predictor.blocks[1][2].parameters())
    net.zero_grad(set_to_none=True)
    singles = []
    for e in ex:
        net.zero_grad(set_to_none=True)
        net(e)["loss"].mean().backward()
        singles.append(w.grad.clone())
    net.zero_grad(set_to_none=True)
    for e in ex:
        net(e)["loss"].mean().backward()
    assert torch.equal(w.grad, singles[0] + singles[1])
    # a second batch size gets its own graphs and the first one's come back without a new capture
    st = headgraph._STATE[net.odom_predictor]
    g_two = st.graph
    one = workload.make_example(net, [pool[4]])
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        net(one)["loss"].mean().backward()
    assert st.graph is not None and st.graph is not g_two and st.parked is not None and st.parked[1] is g_two
    net.zero_grad(set_to_none=True)
    net(ex[0])["loss"].mean().backward()
    assert st.graph is g_two and torch.equal(w.grad, singles[0])
    # two forwards BEFORE one backward of the summed losses: the second forward must not overwrite the static activations the
    # first one's backward needs (it takes the eager pass)
    monkeypatch.setattr(headgraph, "MODE", "fwd")
    net.zero_grad(set_to_none=True)
    l1 = net(ex[0])["loss"].mean()
    l2 = net(ex[1])["loss"].mean()
    (l1 + l2).backward()
    assert torch.equal(w.grad, singles[0] + singles[1])
    # mode "fwd" on a non-default stream (capture on the caller's stream, weight gradients on the leaf stream): two backward
    # passes WITHOUT a reset -- the add into the gradient that is already there must wait for the leaf stream (the alias
    # leaves never carry a gradient, so streams.leaf() cannot see the accumulation by itself)
    torch.cuda.synchronize()
    headgraph._STATE.pop(net.odom_predictor, None)      # the graphs captured from the legacy default stream above
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(2):      # the stream's own capture of this shape (first call warm-up, second captures)
            net.zero_grad(set_to_none=True)
            net(ex[0])["loss"].mean().backward()
        assert headgraph._STATE[net.odom_predictor].graph.on_callers_stream
        for rep in range(3):
            net.zero_grad(set_to_none=True)
            for e in ex:
                net(e)["loss"].mean().backward()
            assert torch.equal(w.grad, singles[0] + singles[1]), rep
    torch.cuda.current_stream().wait_stream(s_)


def test_deferred_weight_gradient_reduces_keep_the_bits(hip, monkeypatch):
    """Round 6: the partials -> gradient stage of all 70 weight gradients of a step runs as ONE launch per stream at the end of
    the backward pass (rslo_amd/streams.py DEFER_REDUCES, csrc/wgrad_reduce.hip) instead of a launch per layer.  Six optimizer
    steps end in the same weights / statistics / losses either way; gradient accumulation over two backward passes without a
    reset (the case where a gradient IS read inside the pass) gives the sum of the two passes' gradients."""
    import hashlib
    from rslo_amd import headgraph, streams
    pool = [list(reduced_pair(i)[:2]) for i in range(6)]

    def run(defer, graph):
        monkeypatch.setattr(streams, "DEFER_REDUCES", defer)
        monkeypatch.setattr(headgraph, "ENABLED", bool(graph))
        monkeypatch.setattr(headgraph, "MODE", graph)
        torch.manual_seed(7)
        net, _ = workload.build_network()
        net.train()
        losses = real_training_steps(net, 6, lambda i: [pool[(2 * i) % 6], pool[(2 * i + 1) % 6]])
        torch.cuda.synchronize()
        state = torch.cat([t.detach().double().reshape(-1) for t in list(net.parameters()) + list(net.buffers())])
        return hashlib.sha256(state.cpu().numpy().tobytes()).hexdigest(), losses, net
    h0, l0, _ = run(False, None)
    for graph in (None, "fwd"):
        h1, l1, net = run(True, graph)
        assert l0 == l1 and h0 == h1, graph
    # accumulation: the second pass must not defer (AccumulateGrad adds into the gradient that is already there)
    ex = [workload.make_example(net, [pool[0], pool[1]]), workload.make_example(net, [pool[2], pool[3]])]
    probes = [next(net.odom_predictor.blocks[1][2].parameters()), net.middle_feature_extractor.middle_conv[3].weight,
              net.middle_feature_extractor.middle_conv_tail[0].weight]
    singles = []
    for e in ex:
        net.zero_grad(set_to_none=True)
        net(e)["loss"].mean().backward()
        singles.append([p.grad.clone() for p in probes])
    net.zero_grad(set_to_none=True)
    for e in ex:
        net(e)["loss"].mean().backward()
    for p, a, b in zip(probes, singles[0], singles[1]):
        assert torch.equal(p.grad, a + b)


def test_head_graph_replays_the_multi_rank_syncbn_path_bit_identically(hip, monkeypatch):
    """Round 6: the step a rank of an N > 1 job runs.  A one-rank process group with the multi-rank SyncBatchNorm path forced
    (RSLO_FORCE_SYNCBN_PATH=1: every layer's statistics go through the node's peer comm -- inside the BatchNorm kernel on the
    register-cached maps, statistics -> exchange kernel -> apply on the 96x176 maps).  With the head's forward replayed from a
    hipGraph (the exchanges carry numbers relative to a device word that is set in front of every replay) ten optimizer steps end
    in the same weights, statistics and losses as the same ten steps issued launch by launch."""
    import hashlib
    import socket
    import torch.distributed as dist
    import apex.parallel as AP
    from rslo_amd import headgraph, peer
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    monkeypatch.setenv("RSLO_FORCE_SYNCBN_PATH", "1")
    monkeypatch.setattr(AP, "FORCE_MULTI", True)
    dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
    pool = [list(reduced_pair(i)[:2]) for i in range(6)]
    try:
        def run(mode):
            monkeypatch.setattr(headgraph, "ENABLED", bool(mode))
            monkeypatch.setattr(headgraph, "MODE", mode)
            torch.manual_seed(7)
            net, _ = workload.build_network()
            net.train()
            losses = real_training_steps(net, 10, lambda i: [pool[(2 * i) % 6], pool[(2 * i + 1) % 6]])
            torch.cuda.synchronize()
            peer.check_all()
            state = torch.cat([t.detach().double().reshape(-1) for t in list(net.parameters()) + list(net.buffers())])
            return hashlib.sha256(state.cpu().numpy().tobytes()).hexdigest(), losses, headgraph._STATE.get(net.odom_predictor)
        h0, l0, st0 = run(None)
        assert st0 is None
        comm = peer.comm_for(None)
        assert comm is not None and comm.world == 1            # the multi-rank path really ran (a world-size-1 peer comm)
        for mode in ("fwd", "full"):
            h1, l1, st1 = run(mode)
            assert st1 is not None and st1.graph is not None and not st1.failed, mode
            assert st1.graph.comm is comm and st1.graph.n_fwd_exchanges >= 45, (mode, st1.graph.n_fwd_exchanges)
            if mode == "full":
                assert st1.graph.n_bwd_exchanges >= 45
            assert l0 == l1 and h0 == h1, mode
    finally:
        peer.shutdown()
        dist.destroy_process_group()


def test_basic_blocks_are_exact_on_the_inputs_they_see_in_the_network(hip):
    """The BEV encoder's BasicBlocks, each evaluated STAND-ALONE on the input and output gradient it receives inside the
    network (default init, sparse reduced-ring BEV map, warm-up regime -- the state in which whole-network gradient
    comparisons are ill-conditioned, see test_parity_on_weights_produced_by_real_optimizer_steps): the fused HIP node
    (conv3x3 -> SyncBN -> ReLU -> conv3x3 -> SyncBN -> + x -> ReLU, forward and backward) against the plain torch formulation
    in float64 on the same inputs.  Input gradient, both weight gradients and the four affine gradients to 2e-5."""
    import torch.nn.functional as F
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(50)
    ex = workload.make_example(net, [list(reduced_pair(1)[:2])])
    picks = {(0, 1): {}, (1, 1): {}, (1, 2): {}, (2, 2): {}}
    for (si, bi), cap in picks.items():
        blk = net.odom_predictor.blocks[si][bi]
        assert blk.downsample is None

        def fwd(x, _orig=blk.forward, _cap=cap):
            _cap["x"] = (x[0] if isinstance(x, (list, tuple)) else x).detach().clone()
            out = _orig(x)
            t = out[0] if isinstance(out, (list, tuple)) else out
            t.register_hook(lambda g_: _cap.__setitem__("gy", g_.detach().clone()))
            return out
        blk.forward = fwd
    net(ex)["loss"].backward()
    for (si, bi), cap in picks.items():
        blk = net.odom_predictor.blocks[si][bi]
        del blk.forward
        x, gy = cap["x"], cap["gy"]
        c1, c2 = getattr(blk.conv1, "conv1", blk.conv1), getattr(blk.conv2, "conv1", blk.conv2)
        f = lambda t: t.detach().cpu().double().requires_grad_(True)      # noqa: E731
        xr, w1, w2, g1, b1, g2, b2 = (f(t) for t in (x, c1.weight, c2.weight, blk.bn1.weight, blk.bn1.bias, blk.bn2.weight,
                                                     blk.bn2.bias))
        y1 = F.relu(F.batch_norm(F.conv2d(xr, w1, None, 1, 1), None, None, g1, b1, True, 0.0, blk.bn1.eps))
        y2 = F.relu(F.batch_norm(F.conv2d(y1, w2, None, 1, 1), None, None, g2, b2, True, 0.0, blk.bn2.eps) + xr)
        y2.backward(gy.cpu().double())
        b = copy.deepcopy(blk)
        b.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        out = b([xg, None])
        yg = out[0] if isinstance(out, (list, tuple)) else out
        assert type(yg.grad_fn).__name__ == "_BasicBlockFnBackward"
        yg.backward(gy)
        got = [yg, xg.grad, getattr(b.conv1, "conv1", b.conv1).weight.grad, getattr(b.conv2, "conv1", b.conv2).weight.grad,
               b.bn1.weight.grad, b.bn1.bias.grad, b.bn2.weight.grad, b.bn2.bias.grad]
        want = [y2, xr.grad, w1.grad, w2.grad, g1.grad, b1.grad, g2.grad, b2.grad]
        errs = [rel(a_, w_) for a_, w_ in zip(got, want)]
        print("block %d.%d stand-alone vs float64 (y, dx, dw1, dw2, dg1, db1, dg2, db2):" % (si, bi), " ".join("%.1e" % e for e in errs))
        assert max(errs) < 2e-5, ((si, bi), errs)


def test_amp_o1_bf16_step_tracks_the_fp32_step(hip):
    """BASELINE config C4, per-GPU part: apex.amp.initialize(..., "O1") switches the 32/64-channel encoder trunk to
    bf16 rows and the dense 3x3 convolutions of the head to bf16 operands (fp32 accumulation, fp32 master weights; the
    covariance branch, BatchNorm, vote and the loss stay fp32).  Stated tolerance against the fp32 step on the same
    inputs and weights: pose 2e-2 of the largest component, loss terms 5e-2, and every parameter gradient that is not
    analytically zero points the same way (cosine to the fp32 gradient: median > 0.9, every tensor > 0.6; measured
    median 0.957, minimum 0.80 -- the consistency loss turns a 1e-3 pose change into a visible change of direction, see
    check_three_way).  bf16 carries 8 significant bits; the kernels themselves are pinned on rounded operands in
    test_gpu_kernels.py."""
    from apex import amp
    from rslo_amd import precision
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    trained_like_init(net)
    ex = workload.make_example(net, [list(reduced_pair(1)[:2])])
    net16 = copy.deepcopy(net)
    ret = net(ex)
    ret["loss"].backward()
    opt = torch.optim.SGD(net16.parameters(), lr=0.0)
    try:
        net16, opt = amp.initialize(net16, opt, opt_level="O1")
        assert precision.low_precision() is torch.bfloat16 and amp.state_dict()["loss_scale"] == 1.0
        seen = {}
        net16.middle_feature_extractor.middle_conv_tail[0].register_forward_hook(
            lambda m, i, o: seen.update(trunk=o.features.dtype))
        net16.middle_feature_extractor.middle_cov_deconv[0].register_forward_hook(
            lambda m, i, o: seen.update(cov=o.features.dtype))
        ret16 = net16(ex)
        with amp.scale_loss(ret16["loss"], opt) as scaled:
            assert scaled is ret16["loss"]
            scaled.backward()
    finally:
        amp.initialize(net16, opt, opt_level="O0")
    assert precision.low_precision() is None
    assert seen == {"trunk": torch.bfloat16, "cov": torch.float32}
    for k in ("translation_preds", "rotation_preds"):
        assert ret16[k].dtype == torch.float32 and rel(ret16[k], ret[k]) < 2e-2, k
    for k in ("loss", "translation_loss", "pyramid_loss", "C_loss"):
        assert rel(ret16[k], ret[k]) < 5e-2, k
    cos = []
    skip = bias_before_bn(net)
    for (n, p), (_, q) in zip(net.named_parameters(), net16.named_parameters()):
        if n in skip or p.grad is None or float(p.grad.abs().max()) < 1e-6:
            continue
        assert q.grad is not None and q.grad.dtype == torch.float32, n
        cos.append(float(torch.nn.functional.cosine_similarity(p.grad.flatten().double(), q.grad.flatten().double(), dim=0)))
    cos = np.array(cos)
    assert len(cos) >= 170 and np.median(cos) > 0.9 and cos.min() > 0.6, (np.median(cos), cos.min())


def _probe_step(net, ex):
    """Forward through encoder + head, the loss values, and the backward of a seeded LINEAR functional of every output
    the loss consumes (poses, both confidence maps, the pyramid maps, the covariance head).  The training loss itself
    turns a 1e-7 pose difference into ~1e-3 of gradient (check_three_way); at bf16 operand precision poses agree to
    ~5e-5, so its gradients say nothing about the kernels -- a linear functional exercises every backward kernel of
    the step with a conditioning of one."""
    B = ex["num_voxels"][0].shape[0]
    preds = net.network_forward(ex["voxels"], ex["num_points"], ex["coordinates"], B, example=ex)
    gen = torch.Generator().manual_seed(5)
    hw = preds["t_conf"].shape[-1] * preds["t_conf"].shape[-2]
    outs = [preds["translation_preds"][0], preds["rotation_preds"][0], preds["t_conf"] * hw, preds["r_conf"] * hw]
    outs += [pm[0] for pm in preds["pyramid_motion"]] + list(preds["middle_conf_preds"])
    probe = 0.0
    for o in outs:
        w = torch.randn(o.shape, generator=gen).to(device=o.device, dtype=o.dtype)
        probe = probe + (o * w).sum() / float(o.numel()) ** 0.5
    vals = net.loss(ex, preds)
    probe.backward()
    vals["_probe_outputs"] = [o.detach() for o in outs]
    return vals, probe.detach()


def _c4_step_vs_arbiter(pairs):
    """One apex-O1 (bf16 operand) step on the GPU and the SAME step on the CPU with the SAME rounding points
    (oracle/cpu_backend.py: bf16 rows and operands, float64 accumulation, float64 everything else).  Returns
    (gpu loss outputs, arbiter loss outputs, (gpu probe, arbiter probe), [(probe-gradient error vs arbiter, name)])."""
    from apex import amp
    from rslo_amd import precision
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    trained_like_init(net)
    ex = workload.make_example(net, pairs)
    net_f64 = copy.deepcopy(net).cpu().double()
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    try:
        net, opt = amp.initialize(net, opt, opt_level="O1")
        assert precision.low_precision() is torch.bfloat16
        ret, probe = _probe_step(net, ex)
        cpu_backend.result_dtype = torch.float64
        with cpu_backend.patched():
            ret64, probe64 = _probe_step(net_f64, example_to_f64(example_to_cpu(ex)))
    finally:
        cpu_backend.result_dtype = torch.float32
        amp.initialize(net, opt, opt_level="O0")
    rows = []
    skip = bias_before_bn(net)
    for (n, p), (_, q) in zip(net.named_parameters(), net_f64.named_parameters()):
        if n in skip or q.grad is None or float(q.grad.abs().max()) < 1e-6:
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32, n
        cos = float(torch.nn.functional.cosine_similarity(p.grad.flatten().double().cpu(), q.grad.flatten(), dim=0))
        rows.append((rel(p.grad, q.grad), cos, n))
    return ret, ret64, (probe, probe64), rows


def _check_c4(tag, ret, ret64, probes, rows, pose_bar, loss_bar, out_bar, cos_median_bar, cos_min_bar):
    e = np.array([r[0] for r in rows])
    cos = np.array([r[1] for r in rows])
    outs = [rel(a, b) for a, b in zip(ret["_probe_outputs"], ret64["_probe_outputs"])]
    print("C4 %s: pose" % tag, [rel(ret[k], ret64[k]) for k in ("translation_preds", "rotation_preds")], "loss",
          [rel(ret[k], ret64[k]) for k in ("loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss")],
          "outputs", outs, "probe-gradient rel median", float(np.median(e)), "cos median / min", float(np.median(cos)),
          min((r[1], r[2]) for r in rows), "tensors", len(rows))
    for k in ("translation_preds", "rotation_preds"):
        assert ret[k].dtype == torch.float32 and rel(ret[k], ret64[k]) < pose_bar, (k, rel(ret[k], ret64[k]))
    for k in ("loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
        assert rel(ret[k], ret64[k]) < loss_bar, (k, rel(ret[k], ret64[k]))
    assert max(outs) < out_bar, outs
    assert len(rows) >= 170
    assert np.median(cos) > cos_median_bar and cos.min() > cos_min_bar, (np.median(cos), min((r[1], r[2]) for r in rows))


def test_c4_bf16_step_matches_arbiter_on_rounded_operands(hip):
    """BASELINE config C4, per-GPU part, reduced pair.  The reference under apex O1 keeps the BEV head, the loss, SVD,
    chamfer in fp32 islands and runs the sparse trunk in half precision (train_hdf5.py:456-463);
    here: bf16 rows in the 32/64-channel trunk, bf16 operands in the dense 3x3 layers, everything else fp32.  The bar
    is parity with an arbiter that rounds at the SAME points and accumulates in float64 -- not "tracks the fp32 step":
    what is left is fp32-vs-float64 accumulation plus the bf16 roundings that flip when a sum lands on a tie."""
    _check_c4("reduced", *_c4_step_vs_arbiter([list(reduced_pair(1)[:2])]), *C4_BARS)


def test_c4_full_size_bf16_step_matches_arbiter_on_rounded_operands(hip):
    """The same at BASELINE's size: bs 4 frame pairs of full 64-ring scans (~250 k voxels in the batched trunk)."""
    _check_c4("full", *_c4_step_vs_arbiter([list(reduced_pair(b + 1, rings=64)[:2]) for b in range(4)]), *C4_BARS)


# Stated C4 tolerances against the arbiter: pose 2e-3 of the largest component (measured 5.8e-5 reduced / 4.1e-4 full size),
# loss terms 5e-3 (1.0e-3 / 1.2e-3), every output map the loss consumes 8e-2 of its largest entry (pose maps 4e-4,
# confidences 5e-3..8e-3, the two pyramid heads 3.8e-2 / 4.2e-2), probe-gradient cosine median > 0.9 (0.963 / 0.960) and
# > 0.75 for every tensor (0.91 / 0.90).  Why not tighter: GPU and arbiter differ by accumulation width only, but a 1e-7
# difference flips the bf16 rounding of ~3e-4 of the next layer's operands (1 bf16 ulp = 4e-3 each), and the random-init
# head amplifies perturbations ~1e3-fold on the way to the gradients (BatchNorm over near-constant channels of the
# pyramid heads, eps 1e-3): in fp32 mode the same probe gives a median gradient error of 1.7e-4 for 1e-7 per-layer
# rounding (scripts/c4_diag.py); here the per-layer perturbation is the ~1e-4 of flipped roundings.  The kernels
# themselves are exact to the output's final rounding on identical operands (tests/test_gpu_kernels.py).
C4_BARS = (2e-3, 5e-3, 8e-2, 0.9, 0.75)


def test_eval_forward_sees_weights_written_through_data(hip):
    """In-place writes through `.data` do not bump a parameter's version counter; the cached split-bf16 operands of
    the sparse convolutions must not survive them."""
    torch.manual_seed(3)
    net, _ = workload.build_network()
    net.train()
    ex = workload.make_example(net, [list(reduced_pair(2)[:2])])
    net(ex)["loss"].backward()                      # training forward: operands cached on the parameters
    net.eval()
    with torch.no_grad():
        before = net(ex)["translation_preds"].clone()
        for p in net.middle_feature_extractor.parameters():
            p.data.mul_(0.5)                        # e.g. an EMA swap
        after = net(ex)["translation_preds"].clone()
        fresh = copy.deepcopy(net)                  # no cached operands at all
        for p in fresh.parameters():
            p.__dict__.pop("_hip_split", None)
        want = fresh(ex)["translation_preds"]
    assert rel(after, before) > 1e-3          # the halved encoder weights are seen ...
    assert rel(after, want) < 1e-5            # ... exactly as a model without any cached operands sees them


def test_eval_forward_batched_equals_per_sample(hip):
    """bs > 1 is an extension of the reference (middle.py:221 asserts bs == 1): the batched result must equal
    the per-sample results.  Eval mode (BN uses running statistics), fresh BN buffers."""
    torch.manual_seed(3)
    net, _ = workload.build_network()
    net.eval()
    a = reduced_pair(2)
    b = reduced_pair(3)
    with torch.no_grad():
        both = net(workload.make_example(net, [[a[0], a[1]], [b[0], b[1]]]))
        one_a = net(workload.make_example(net, [[a[0], a[1]]]))
        one_b = net(workload.make_example(net, [[b[0], b[1]]]))
    t = torch.cat([one_a["translation_preds"], one_b["translation_preds"]])
    r = torch.cat([one_a["rotation_preds"], one_b["rotation_preds"]])
    assert both["translation_preds"].shape == (2, 3) and both["rotation_preds"].shape == (2, 4)
    assert rel(both["translation_preds"], t) < 1e-4 and rel(both["rotation_preds"], r) < 1e-4


def test_spconv_api_indice_pairs_view(hip):
    """The spconv-1.x `indice_pairs` view of our tables equals the oracle's export (bit-exact indices)."""
    import oracle as O
    import spconv
    rng = np.random.default_rng(0)
    dims = [9, 20, 18]
    lin = rng.choice(2 * 9 * 20 * 18, size=1200, replace=False)
    x_ = lin % 18; r = lin // 18
    y_ = r % 20; r //= 20
    z_ = r % 9; b_ = r // 9
    coords = np.stack([b_, z_, y_, x_], 1).astype(np.int32)
    coords = coords[np.argsort(coords[:, 0], kind="stable")]
    feats = torch.randn(len(coords), 16, device="cuda")
    t = spconv.SparseConvTensor(feats, torch.from_numpy(coords).cuda(), dims, 2)
    conv = spconv.SparseConv3d(16, 32, 3, 2, padding=1, indice_key="c").cuda()
    sub = spconv.SubMConv3d(16, 16, 3, indice_key="s").cuda()
    y = conv(sub(t))
    oc, od, nbr, nbrT = O.rulebook_conv(coords, 2, dims, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert y.spatial_shape == od and (y.indices.cpu().numpy() == oc).all()
    pairs, num = t.indice_dict["c"].indice_pairs()
    opairs, onum = O.pairs_from_nbrT(nbrT)
    assert (num.cpu().numpy() == onum).all() and (pairs.cpu().numpy() == opairs).all()
    spairs, snum = t.indice_dict["s"].indice_pairs()
    osub = O.rulebook_subm(coords, 2, dims)
    op2, on2 = O.pairs_from_nbrT(osub[:, ::-1].copy())
    assert (snum.cpu().numpy() == on2).all() and (spairs.cpu().numpy() == op2).all()


def test_training_batched_loss_equals_per_sample(hip):
    """bs 2 training step: the padded ragged loss batch must give the per-sample consistency / ICP results.
    BN layers are put in eval mode so that the only coupling between samples would be a batching error."""
    torch.manual_seed(5)
    net, _ = workload.build_network()
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    net.global_step.fill_(2000)
    trained_like_init(net)
    a, b = reduced_pair(4), reduced_pair(5, rings=32)      # different sizes -> ragged batch
    both = net(workload.make_example(net, [[a[0], a[1]], [b[0], b[1]]]))
    ra = net(workload.make_example(net, [[a[0], a[1]]]))
    rb = net(workload.make_example(net, [[b[0], b[1]]]))
    # C_loss = exp(-alpha) * mean over pairs + alpha with alpha = 0
    assert rel(both["C_loss"], (ra["C_loss"] + rb["C_loss"]) / 2) < 1e-4
    assert rel(both["translation_preds"], torch.cat([ra["translation_preds"], rb["translation_preds"]])) < 1e-4
    assert rel(both["translation_loss"], (ra["translation_loss"] + rb["translation_loss"]) / 2) < 1e-3



def test_prefetched_example_equals_inline(hip):
    """ExamplePrefetcher (voxelization + rulebook planning on a side stream / helper thread) hands the network the
    same example as the in-line path: identical loss and poses over consecutive steps with different clouds."""
    torch.manual_seed(9)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    pairs = [reduced_pair(7), reduced_pair(8, rings=32), reduced_pair(9)]
    batches = [[[torch.from_numpy(p[0]).cuda(), torch.from_numpy(p[1]).cuda()]] for p in pairs]
    ref = []
    for clouds in batches:
        r = net(workload.make_example(net, clouds))
        r["loss"].mean().backward()
        ref.append((r["loss"].detach().clone(), r["translation_preds"].clone(),
                    torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None]).clone()))
        net.zero_grad(set_to_none=True)
    # BN running statistics moved during the reference pass: start the second pass from the same state
    torch.manual_seed(9)
    net2, _ = workload.build_network()
    net2.train()
    net2.global_step.fill_(2000)
    pf = workload.ExamplePrefetcher(net2)
    pf.submit(batches[0])
    for i, clouds in enumerate(batches):
        ex = pf.get()
        assert "sparse_plan" in ex
        inline = workload.make_example(net2, clouds)        # the example itself is bit-identical to the in-line one
        for key in ("voxels", "coordinates", "num_points"):
            assert len(ex[key]) == len(inline[key])
            for a, b in zip(ex[key], inline[key]):
                assert torch.equal(a, b), key
        r = net2(ex)
        if i + 1 < len(batches):
            pf.submit(batches[i + 1])
        r["loss"].mean().backward()
        g = torch.cat([p.grad.reshape(-1) for p in net2.parameters() if p.grad is not None])
        # two runs of the SAME path already differ at the 1e-6 level (library BN/conv reductions are not run-to-run
        # bit-stable) and a nearest-neighbour or ROI-threshold tie that flips moves the loss by ~2e-5
        # (scripts/determinism.py shows the same spread with and without the prefetcher)
        # -- so the numeric comparison only guards against a wrong example, not against those flips
        assert rel(r["loss"], ref[i][0]) < 5e-4 and rel(r["translation_preds"], ref[i][1]) < 5e-4
        assert rel(g, ref[i][2]) < 5e-2
        net2.zero_grad(set_to_none=True)
    pf.close()


def _plan_tables(plan):
    """Every table of a planned SparseConvTensor, keyed by indice_key."""
    out = {"coords": plan.indices}
    for key, rb in plan.indice_dict.items():
        out[key + ".nbr"] = rb.nbr
        if rb.nbrT is not None:
            out[key + ".nbrT"] = rb.nbrT
            out[key + ".out_coords"] = rb.out_index.coords
            o = rb.order("nbrT")
            if o is not None:
                out[key + ".order"] = o
        pin, pout, koff = rb.pairs()
        npairs = int(koff[-1])
        out[key + ".pin"], out[key + ".pout"], out[key + ".koff"] = pin[:npairs], pout[:npairs], koff
    return out


@pytest.mark.parametrize("case", ["pair", "bs2_ragged", "empty_cloud"])
def test_native_plan_equals_python_plan(hip, case):
    """rslo_plan_encoder (one foreign call, row counts on the device, capacity-sized arena) produces the example dict
    and EVERY table of the Python-issued plan bit for bit: voxels / coordinates / point counts per frame, site order of
    every level, SubM and strided tables, transposed tables, tile orders, pair lists, per-level batch offsets."""
    from rslo_amd import plan as native_plan
    torch.manual_seed(3)
    net, _ = workload.build_network()
    net.train()
    if case == "pair":
        pairs = [reduced_pair(7, rings=32)]
    elif case == "bs2_ragged":
        pairs = [reduced_pair(8), reduced_pair(9, rings=32)]
    else:      # a sample whose second frame has no point inside the range
        p = reduced_pair(10)
        far = p[1].copy()
        far[:, :3] += 1000.0
        pairs = [reduced_pair(11), (p[0], far)]
    clouds = [[torch.from_numpy(c).cuda() for c in pr[:2]] for pr in pairs]
    ref = net.plan_example(workload.make_example(net, clouds))
    planner = native_plan.EncoderPlanner(net, synthetic.MAX_VOXELS)
    for _ in range(2):          # second round: the arena ring hands out fresh memory, results must not depend on it
        ex = planner.finish(planner.submit(clouds))
    assert planner.fallbacks == 0
    for key in ("voxels", "coordinates", "num_points", "num_voxels"):
        assert len(ex[key]) == len(ref[key])
        for a, b in zip(ex[key], ref[key]):
            assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.cpu(), b.cpu()), key
    ta, tb = _plan_tables(ex["sparse_plan"]), _plan_tables(ref["sparse_plan"])
    assert sorted(ta) == sorted(tb)
    for k in ta:
        assert ta[k].shape == tb[k].shape and torch.equal(ta[k], tb[k]), k
    pa, pb = ex["sparse_plan"], ref["sparse_plan"]
    assert pa.batch_offsets() == pb.batch_offsets()
    assert torch.equal(pa.site_index().batch_offs_dev, pb.site_index().batch_offs_dev)
    if case == "empty_cloud":       # the loss has no defined value for a frame without voxels
        return
    # and the step on it: same loss / poses as on the Python plan
    net.global_step.fill_(2000)
    r0 = net(ref)
    r1 = net(ex)
    assert rel(r1["loss"], r0["loss"]) < 5e-4 and rel(r1["translation_preds"], r0["translation_preds"]) < 5e-4
    r1["loss"].mean().backward()


def test_native_plan_overflow_falls_back_to_exact_sizes(hip):
    """A strided level with MORE sites than its input (isolated voxels: each feeds up to 8 outputs of a k3 s2 conv)
    overruns the default capacity; the planner sees the flag and re-plans with exact sizes instead of truncating."""
    from rslo_amd import plan as native_plan
    net, _ = workload.build_network()
    g = np.random.default_rng(0)
    n = 3000       # isolated points with odd voxel coordinates on a coarse lattice
    ix, iy, iz = g.integers(0, 700, n) * 2 + 1, g.integers(0, 380, n) * 2 + 1, g.integers(0, 19, n) * 2 + 1
    xyz = np.stack([ix * 0.1 - 70.4 + 0.05, iy * 0.1 - 38.4 + 0.05, iz * 0.2 - 3 + 0.1], 1).astype(np.float32)
    pts = np.concatenate([xyz, np.zeros((n, 4), np.float32)], 1)
    pts[:, 6] = 1.0
    clouds = [[torch.from_numpy(pts).cuda(), torch.from_numpy(pts[::2].copy()).cuda()]]
    planner = native_plan.EncoderPlanner(net, synthetic.MAX_VOXELS)
    ex = planner.finish(planner.submit(clouds))
    assert planner.fallbacks == 1
    ref = net.plan_example(workload.make_example(net, clouds))
    ta, tb = _plan_tables(ex["sparse_plan"]), _plan_tables(ref["sparse_plan"])
    for k in ta:
        assert torch.equal(ta[k], tb[k]), k
    assert ref["sparse_plan"].indice_dict["conv3d2"].nbr.shape[0] > ref["sparse_plan"].indices.shape[0]


def test_short_training_run_stays_finite(hip):
    """End-to-end behaviour of the assembled step (encoder + head + loss + the reference's optimizer wrapper and
    OneCycle schedule): 25 optimizer steps on one reduced synthetic pair keep the loss and every parameter finite and
    move the weights.  (No monotone-decrease claim: from a random initialisation at step 2000 -- no warm-up, the
    predicted pose drives the association -- the reference's self-supervised loss is chaotic for the first steps.)"""
    from rslo.builder import lr_scheduler_builder, optimizer_builder
    from rslo.utils import config_text
    torch.manual_seed(3)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    cfg = config_text.shipped_config().train_config
    opt = optimizer_builder.build(cfg.optimizer, net)
    sched = lr_scheduler_builder.build(cfg.optimizer, opt, 60)        # a 60-step cycle: the peak lr is reached at step 3
    a = reduced_pair(11)
    clouds = [[torch.from_numpy(a[0]).cuda(), torch.from_numpy(a[1]).cuda()]]
    params = [p for p in net.parameters() if p.requires_grad]
    before = torch.cat([p.detach().reshape(-1) for p in params]).clone()
    losses = []
    for it in range(25):
        sched.step(it)
        opt.zero_grad()
        ret = net(workload.make_example(net, clouds))
        ret["loss"].mean().backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
        losses.append(float(ret["loss"]))
    assert all(np.isfinite(losses))
    assert all(bool(torch.isfinite(p).all()) for p in params)
    assert max(losses) < 1e6, losses
    after = torch.cat([p.detach().reshape(-1) for p in params])
    assert float((after - before).abs().max()) > 1e-4


@pytest.mark.parametrize("step", [2000, 100])
def test_fused_loss_tail_matches_torch_formulation(hip, step):
    """rslo_loss_tail_fwd / _bwd (pose L2 terms, pyramid reduction, consistency reduce, loss weights, total: one launch
    each way) against the op-by-op torch formulation that the reference-generated goldens pin (check_create_loss): every
    loss term and every parameter gradient of a training step, after warm-up (step 2000) and inside it (step 100:
    identity pose in the consistency loss, 5 ICP rounds)."""
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(step)
    trained_like_init(net)
    with torch.no_grad():       # distinct log-variances: their gradients must land on the right modules
        net._translation_loss.alpha.fill_(0.3)
        net._rotation_loss.alpha.fill_(-2.1)
        net._consistency_loss.alpha.fill_(0.7)
    ex = workload.make_example(net, [list(reduced_pair(1)[:2]), list(reduced_pair(2)[:2])])
    outs = []
    for fused in (True, False):
        net.fused_loss_tail = fused
        net.zero_grad(set_to_none=True)
        ret = net(ex)
        ret["loss"].mean().backward()
        outs.append((ret, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    (ra, ga), (rb, gb) = outs
    for k in ("loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
        assert ra[k].shape == rb[k].shape == (1,) and rel(ra[k], rb[k]) < 2e-6, (k, float(ra[k]), float(rb[k]))
    assert sorted(ga) == sorted(gb) and len(ga) >= 200
    skip = bias_before_bn(net)                  # analytically zero gradients: rounding noise on both sides
    for n in gb:
        if n not in skip and float(gb[n].abs().max()) > 1e-5:
            assert rel(ga[n], gb[n]) < 5e-5, n      # same kernels upstream; the tail's sums are in a different order
    for n in ("_translation_loss.alpha", "_rotation_loss.alpha"):
        assert n in ga and float(ga[n].abs().max()) > 0


def test_fused_head_tail_matches_torch_formulation(hip):
    """csrc/headtail.hip (quaternion normalisation, the four masked softmaxes, mask / weight pyramid, masked maps:
    3 launches each way) against the torch ops of the reference formulation (odom_pred.py:227-264, confidence.py:26-34)
    inside a full training step: every map the head returns, the loss terms and every parameter gradient."""
    torch.manual_seed(7)
    net, _ = workload.build_network()
    net.train()
    net.global_step.fill_(2000)
    trained_like_init(net)
    ex = workload.make_example(net, [list(reduced_pair(1)[:2]), list(reduced_pair(2)[:2])])
    head = net.odom_predictor
    outs = []
    for fused in (True, False):
        head._fused_tail_ok(torch.zeros(1, device="cuda"), torch.zeros(1, 1, 8, 8, device="cuda"))    # fills the static flag
        assert head.__dict__["_fused_tail_static"] is True
        if not fused:
            head.__dict__["_fused_tail_static"] = False
        net.zero_grad(set_to_none=True)
        ret = net(ex)
        ret["loss"].mean().backward()
        outs.append((ret, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
        head.__dict__.pop("_fused_tail_static")
    (ra, ga), (rb, gb) = outs
    for k in ("t_conf", "r_conf", "tq_map_g"):
        assert ra[k].shape == rb[k].shape and rel(ra[k], rb[k]) < 1e-5, k
    assert len(ra["pyramid_motion"]) == len(rb["pyramid_motion"]) == 3
    for (pa, wa), (pb, wb) in zip(ra["pyramid_motion"], rb["pyramid_motion"]):
        assert pa.shape == pb.shape and wa.shape == wb.shape
        assert rel(pa, pb) < 1e-5 and rel(wa, wb) < 1e-5
    for k in ("translation_preds", "rotation_preds"):
        assert rel(ra[k], rb[k]) < 1e-5, k
    for k in ("loss", "translation_loss", "rotation_loss", "pyramid_loss", "C_loss"):
        assert rel(ra[k], rb[k]) < 1e-4, k
    skip = bias_before_bn(net)
    errs = [(rel(ga[n], gb[n]), n) for n in gb if n not in skip and float(gb[n].abs().max()) > 1e-5]
    print("fused head tail: gradient rel error median %.2e max %.2e (%s)" % (np.median([e for e, _ in errs]), *max(errs)))
    assert sorted(ga) == sorted(gb) and np.median([e for e, _ in errs]) < 2e-4 and max(errs)[0] < 2e-2, max(errs)


def test_weight_gradients_on_the_leaf_stream_are_the_same_gradients(hip):
    """rslo_amd.streams.leaf: the dense weight-gradient kernels on a second stream, joined by an
    engine callback at the end of backward.  Same kernels on the same operands: every gradient equals the single-stream
    run to rounding level, over several steps with the optimizer in between (the join must also order the NEXT step's
    writes)."""
    from rslo_amd import streams
    grads = []
    for on in (True, False):
        streams.ENABLED = on
        try:
            torch.manual_seed(7)
            net, _ = workload.build_network()
            net.train()
            net.global_step.fill_(2000)
            trained_like_init(net)
            opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-8)      # smooth dynamics
            ex = workload.make_example(net, [list(reduced_pair(1)[:2]), list(reduced_pair(2)[:2])])
            per_step = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                net(ex)["loss"].mean().backward()
                per_step.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
                opt.step()
            grads.append(per_step)
        finally:
            streams.ENABLED = True
    a, b = grads
    skip = bias_before_bn(net)
    # two runs of one path are not bit-stable (float atomics in the residual backward, the library's atomically reduced
    # stride-2 weight gradient), so "the same" means rounding level on the first step -- a weight gradient read before
    # its kernel finished, or written over by the next step, would be off by O(1) -- and bounded drift afterwards
    for n in a[0]:
        if n not in skip and float(b[0][n].abs().max()) > 1e-5:
            assert rel(a[0][n], b[0][n]) < 2e-4, n
    for k in (1, 2):
        errs = [rel(a[k][n], b[k][n]) for n in a[k] if n not in skip and float(b[k][n].abs().max()) > 1e-5]
        assert np.median(errs) < 2e-2 and max(errs) < 0.2, (k, np.median(errs), max(errs))      # measured 3e-3 / 1e-2


def test_fused_vote_matches_torch_formulation(hip):
    """rslo_vote_fwd/_bwd == from_pointwise_local_transformation_tch + confidence-weighted means (the reference
    formulation, op by op): global map, voted pose, gradients of the local map and of both confidences."""
    torch.manual_seed(13)
    net, _ = workload.build_network()
    head = net.odom_predictor
    B, H, W = 3, 96, 176
    tq = torch.randn(B, 7, H, W, device="cuda")
    tq[:, :3] *= 3.0
    tq[:, 3:] = torch.nn.functional.normalize(tq[:, 3:] + torch.tensor([2.0, 0, 0, 0], device="cuda").view(1, 4, 1, 1), dim=1)
    tq[0, 3:, :4] *= 1.3                                        # not exactly unit: the rotation uses q as given
    tc = torch.softmax(torch.randn(B, 1, H * W, device="cuda"), -1).view(B, 1, H, W)
    rc = torch.softmax(torch.randn(B, 1, H * W, device="cuda") * 2, -1).view(B, 1, H, W)
    go = torch.randn(B, 7, device="cuda")
    res = []
    for fused in (True, False):
        head.fused_vote = fused
        a, b, c = tq.clone().requires_grad_(True), tc.clone().requires_grad_(True), rc.clone().requires_grad_(True)
        tq_g, odom = head.vote(a, b, c)
        (odom * go).sum().backward()
        res.append((tq_g.detach(), odom.detach(), a.grad, b.grad, c.grad))
    head.fused_vote = True
    assert rel(res[0][0], res[1][0]) < 1e-5 and rel(res[0][1], res[1][1]) < 1e-5
    for k in (2, 3, 4):
        assert rel(res[0][k], res[1][k]) < 1e-4, k


def test_three_frame_samples_batched_equals_per_sample(hip):
    """seq_length 3 (the shipped training config): 3 frames -> 3 pairs per sample.  A ragged batch of two such samples
    must reproduce the per-sample consistency loss / poses (BN in eval mode so that samples do not couple)."""
    torch.manual_seed(6)
    net, _ = workload.build_network()
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    net.global_step.fill_(2000)
    trained_like_init(net)
    a0, a1, _ = reduced_pair(21)
    a2 = reduced_pair(22)[1]
    b0, b1, _ = reduced_pair(23, rings=32)
    b2 = reduced_pair(24, rings=32)[1]
    sa, sb = [a0, a1, a2], [b0, b1, b2]
    both = net(workload.make_example(net, [sa, sb]))
    ra = net(workload.make_example(net, [sa]))
    rb = net(workload.make_example(net, [sb]))
    assert both["translation_preds"].shape == (6, 3)
    assert rel(both["translation_preds"], torch.cat([ra["translation_preds"], rb["translation_preds"]])) < 1e-4
    assert rel(both["rotation_preds"], torch.cat([ra["rotation_preds"], rb["rotation_preds"]])) < 1e-4
    assert rel(both["C_loss"], (ra["C_loss"] + rb["C_loss"]) / 2) < 2e-4
    both["loss"].mean().backward()


# ------------------------------------------------------------------------------ registry variants on the device
def test_registry_variants_match_reference_vectors_on_the_gpu(hip):
    """The variants one config flag away from the shipped one (rslo/models/odom_pred.py:319-346 `use_svd` vote,
    rslo/layers/SparseConv.py:222-302 2-D SparseConv, rslo/layers/normalization.py:11-251 MaskSyncBN /
    SemiGlobalSyncBN) against the vectors generated from the reference's own classes (variants_ref.npz), on cuda."""
    check_vote_svd("cuda")
    for mode in (1, 0):
        check_sparse_conv_2d(mode, "cuda")
    for name in ("mask", "semi"):
        check_normalisation_variant(name, "cuda")


@pytest.mark.parametrize("bn_type,use_svd", [("MaskSyncBN", False), ("SemiGlobalSyncBN", False), ("SyncBN", True)])
def test_head_with_registry_variants_steps_on_the_gpu_and_equals_the_cpu_run(hip, bn_type, use_svd):
    """The BEV head with a variant selected, at widths the hand-written dense kernels take (32 / 32 / 64 filters on
    48 x 64 maps: rslo_conv2d_fwd / _wgrad / _s2 underneath), forward + backward on cuda against the SAME head on the
    CPU (plain torch): poses 1e-4, gradient direction of every tensor (cosine > 0.999)."""
    head, xs, out = check_head_variant(bn_type, use_svd, "cuda", widths=(32, 32, 64), up=64, cin=32, hw=(48, 64))
    head_c, xs_c, out_c = check_head_variant(bn_type, use_svd, "cpu", widths=(32, 32, 64), up=64, cin=32, hw=(48, 64))
    for k in ("translation_preds", "rotation_preds"):
        assert rel(out[k][0], out_c[k][0]) < 1e-4, k
    n = 0
    skip = bias_before_bn(head)          # analytically zero gradients: rounding noise on both sides
    for (name, p), (_, q) in zip(head.named_parameters(), head_c.named_parameters()):
        if name in skip or q.grad is None or float(q.grad.abs().max()) < 1e-5:     # (incl. the softmax-shift biases of the
            continue                                                                # confidence heads: analytically zero)
        cos = float(torch.nn.functional.cosine_similarity(p.grad.flatten().cpu().double(), q.grad.flatten().double(), dim=0))
        assert cos > 0.999, (name, cos)
        n += 1
    assert n > 40


def test_encoder_graph_runner_replays_any_scan_bit_equal_to_the_eager_pass(hip):
    """rslo_amd.inference.EncoderGraphRunner (verdict r4 #7): the eval-mode GU encoder of one scan replayed from ONE hipGraph
    per plan arena -- capacity-laid-out plan, rows past each level's device-side count as padding rows
    (rslo_plan_encoder_pad_tails), no host read.  Five DIFFERENT scans through four arenas (so the fifth replays a graph
    captured on another scan): BEV map and the covariance rows of the valid sites equal the eager pass (exact-size plan,
    same modules) bit for bit; the valid-row count on the device equals the eager plan's."""
    import spconv
    from rslo.models import middle
    from rslo_amd import inference, synthetic as S
    from rslo_amd.plan import EncoderPlanner
    gen = spconv.utils.VoxelGenerator(list(S.VOXEL_SIZE), list(S.PC_RANGE), S.MAX_POINTS_PER_VOXEL, S.MAX_VOXELS)
    torch.manual_seed(7)
    enc = middle.get_middle_class("SpMiddleFHDWithCov2_3")(
        [1] + gen.grid_size[::-1].tolist() + [7], bn_type="None", use_leakyReLU=True, num_input_features=7,
        num_filters_down1=[], num_filters_down2=[]).cuda().eval()
    for m in enc.modules():                 # non-trivial running statistics in the covariance branch's BatchNorm1d layers
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)

    class _Net:
        middle_feature_extractor, voxel_generator, training = enc, gen, False
    runner = inference.EncoderGraphRunner(_Net(), S.MAX_VOXELS)
    eager = EncoderPlanner(_Net(), S.MAX_VOXELS)
    clouds = [torch.from_numpy(S.scan(n_el=64 if i != 2 else 32, scan_seed=40 + i)).cuda() for i in range(5)]
    for i, c in enumerate(clouds):
        job = runner.submit(c)
        bev, cov, n_dev = runner.run(job)
        ex = eager.finish(eager.submit([[c]], with_pairs=False))
        vox, num = ex["_frame_major"]
        with torch.no_grad():
            rb, rc = enc(hip.vfe_mean(vox, num), ex["sparse_plan"].indices, 1, plan=ex["sparse_plan"])
        n = int(n_dev.item())
        assert n == rc.shape[0] == vox.shape[0], (i, n, rc.shape)
        assert torch.equal(bev, rb), i
        assert torch.equal(cov[:n], rc), i
    assert len(runner._graphs) == 4         # one capture per arena, the fifth scan replayed the first arena's graph
    # as many handles outstanding as there are arenas: the next plan would overwrite tables a pending replay still reads
    hs = [runner.submit(clouds[0]) for _ in range(4)]
    with pytest.raises(hip.RsloHipError):
        runner.submit(clouds[0])
    for h in hs:
        runner.run(h)
    runner.run(runner.submit(clouds[1]))
    runner.close()
