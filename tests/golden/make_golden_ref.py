"""Generates golden vectors by IMPORTING the Python reference from /root/reference (authoring container
only; the reference never travels to the GPU box -- only the small .npz/.json files written here do).

    python tests/golden/make_golden_ref.py

What gets pinned: the state-dict key/shape list of the shipped network, the BEV head
forward on a reduced configuration, ConfidenceModule, SVDHead, the tq-map local<->global transforms,
AdaptiveWeightedL2Loss, the consistency loss Aleat5_1ChamferL2NormalWeightedALLSVDLoss (values, ICP
outputs, gradients) and the full create_loss assembly.

Shims (documented, minimal): third-party packages that are absent here are replaced by stand-ins --
apex (amp.float_function = identity, SyncBatchNorm = BatchNorm at world size 1), kornia (this repo's
restatement of the 0.4.0 quaternion functions: parity UNPINNED for those two functions), spconv (never
executed: the sparse encoder is not part of these fixtures), numba/h5py/tensorboardX/... (import-only).
The reference's chamfer autograd function refuses CPU tensors (chamfer_distance.py:174-175); it is
routed to the reference's own CPU nnsearch compiled by oracle/build_ref.py.
"""
import collections
import collections.abc
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")


def install_shims():
    for n in ("Iterable", "Mapping", "Sequence"):
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return a[0] if len(a) == 1 and callable(a[0]) else _Anything()

        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Anything()

    class _Loose(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Anything

    def stub(name, **attrs):
        m = _Loose(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    # apex
    amp = stub("apex.amp", float_function=lambda f: f, half_function=lambda f: f,
               register_float_function=lambda *a, **k: None)

    class SyncBatchNorm(torch.nn.BatchNorm2d):
        def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                     process_group=None, channel_last=False, fuse_relu=False):
            super().__init__(num_features, eps, momentum, affine, track_running_stats)

        def _check_input_dim(self, x):
            pass

    par = stub("apex.parallel", SyncBatchNorm=SyncBatchNorm, ReduceOp=None)
    stub("apex.parallel.sync_batchnorm_kernel", SyncBatchnormFunction=None)
    stub("apex", amp=amp, parallel=par)
    # kornia: this repo's restatement
    sys.path.insert(0, os.path.join(ROOT, "rslo_amd", "compat"))
    import kornia  # noqa: F401
    # import-only third parties
    nb = stub("numba", jit=lambda *a, **k: (lambda f: f), njit=lambda *a, **k: (lambda f: f),
              cuda=types.SimpleNamespace(jit=lambda *a, **k: (lambda f: f)))
    nb.__dict__["prange"] = range
    for name in ("transforms3d", "transforms3d.quaternions", "transforms3d.euler", "quaternion", "h5py",
                 "tensorboardX", "cv2", "skimage", "skimage.io", "seaborn", "fire", "open3d", "spconv",
                 "spconv.utils", "matplotlib", "matplotlib.pyplot", "matplotlib.backends",
                 "matplotlib.backends.backend_pdf"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                stub(name)
    sys.modules["tensorboardX"].SummaryWriter = object

    class FakeVG:   # spconv's VoxelGenerator is absent; the builders only need these attributes
        def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels, **kw):
            self.voxel_size = np.array(voxel_size, np.float32)
            self.point_cloud_range = np.array(point_cloud_range, np.float32)

    sys.modules["spconv.utils"].VoxelGenerator = FakeVG
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "rslo"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def ref_chamfer_patch():
    """Route the reference's OneDirectionChamferDistanceWithIdx to its own CPU nnsearch."""
    sys.path.insert(0, ROOT)
    from oracle import build_ref
    build_ref.build(verbose=False)
    cdref = build_ref.load()

    class OneDir(torch.nn.Module):
        def forward(self, xyz1, xyz2):
            a, b = xyz1.detach().contiguous().float(), xyz2.detach().contiguous().float()
            B, N, _ = a.shape
            M = b.shape[1]
            d1, d2 = torch.zeros(B, N), torch.zeros(B, M)
            i1, i2 = torch.zeros(B, N, dtype=torch.int32), torch.zeros(B, M, dtype=torch.int32)
            cdref.forward(a, b, d1, d2, i1, i2)
            return d1, i1

    stub = types.ModuleType("thirdparty.chamfer_distance.chamfer_distance")
    stub.OneDirectionChamferDistanceWithIdx = OneDir
    stub.ChamferDistance = OneDir
    stub.ChamferDistanceWithIdx = OneDir
    pkg = types.ModuleType("thirdparty")
    pkg.__path__ = []
    sub = types.ModuleType("thirdparty.chamfer_distance")
    sub.__path__ = []
    sys.modules["thirdparty"] = pkg
    sys.modules["thirdparty.chamfer_distance"] = sub
    sys.modules["thirdparty.chamfer_distance.chamfer_distance"] = stub


def np_(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def main():
    install_shims()
    ref_chamfer_patch()
    out = {}

    # ---------------------------------------------------------------- pieces
    from rslo.layers.svd import SVDHead
    from rslo.layers.confidence import ConfidenceModule
    from rslo.data.dataset import (from_pointwise_local_transformation_tch,
                                   generate_pointwise_local_transformation_tch)
    import rslo.core.losses as L
    import rslo.utils.pose_utils as PU
    import kornia

    g = torch.Generator().manual_seed(1234)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731

    # SVDHead
    src = rn(2, 3, 300) * 5
    Rt = kornia.quaternion_to_rotation_matrix(torch.tensor([[0.02, -0.01, 0.05, 1.0], [0.0, 0.03, -0.02, 1.0]]))
    tgt = Rt @ src + torch.tensor([[0.5, -0.2, 0.1], [-0.3, 0.4, 0.0]])[..., None] + 0.01 * rn(2, 3, 300)
    w = torch.rand(2, 300, generator=g)
    R, t = SVDHead()(src, tgt, weight=w)
    out.update(svd_src=np_(src), svd_tgt=np_(tgt), svd_w=np_(w), svd_R=np_(R), svd_t=np_(t))

    # quaternion helpers actually in the tree
    q = torch.nn.functional.normalize(rn(50, 4), dim=-1)
    v = rn(50, 3)
    out.update(pu_q=np_(q), pu_v=np_(v), pu_rot=np_(PU.rotate_vec_by_q(v, q)),
               pu_qmult=np_(PU.qmult(q, torch.flip(q, [0]))), pu_qinv=np_(PU.qinv(q)))

    # tq maps (96x176 like the shipped head, and a small one)
    pc_range = np.array([-70.4, -38.4, -3, 70.4, 38.4, 5], np.float32)
    tq = torch.tensor([0.9, 0.03, -0.01, 0.9998, 0.002, -0.003, 0.015])
    tq[3:] = tq[3:] / tq[3:].norm()
    grid = np.array([176, 96, 1])
    vs = (pc_range[3:] - pc_range[:3]) / grid
    origin = ((0 - pc_range[0]) / (pc_range[3] - pc_range[0]) * grid[0],
              (pc_range[4] - 0) / (pc_range[4] - pc_range[1]) * grid[1],
              (0 - pc_range[2]) / (pc_range[5] - pc_range[2]) * grid[2])
    m = generate_pointwise_local_transformation_tch(tq, spatial_size=grid[:2], origin_loc=origin, voxel_size=vs,
                                                    inv_trans_factor=-1)
    back = from_pointwise_local_transformation_tch(m[None], pc_range)
    loc = rn(2, 7, 12, 22)
    glob = from_pointwise_local_transformation_tch(loc, pc_range)
    out.update(tq_vec=np_(tq), tq_map=np_(m), tq_back=np_(back), tq_loc=np_(loc), tq_glob=np_(glob))

    # AdaptiveWeightedL2Loss
    l2 = L.AdaptiveWeightedL2Loss(-2.5, learn_alpha=True, loss_weight=1.0)
    p, tg_, mk = rn(3, 4, 6, 5), rn(3, 4, 6, 5), (torch.rand(3, 1, 6, 5, generator=g) > 0.3).float()
    out.update(l2_pred=np_(p), l2_tgt=np_(tg_), l2_mask=np_(mk), l2_masked=np_(l2(p, tg_, mask=mk)),
               l2_plain=np_(l2(p[:, :, 0, 0], tg_[:, :, 0, 0])))

    # consistency loss: two clouds related by a small motion, covariance params like the encoder's output
    sys.path.insert(0, ROOT)
    from rslo_amd import synthetic
    N = 3000
    base = synthetic.small_cloud(N, seed=3)
    yaw, tr = 0.012, np.array([0.7, 0.05, -0.02], np.float32)
    Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]], np.float32)
    p1 = torch.from_numpy(base[:, :3])[None]
    n1 = torch.from_numpy(base[:, 4:7])[None]
    perm = np.random.default_rng(0).permutation(N)
    p2 = torch.from_numpy(((base[:, :3] - tr) @ Rz)[perm] + 0.01 * np.random.default_rng(1).normal(size=(N, 3)).astype(np.float32))[None].float()
    n2 = torch.from_numpy((base[:, 4:7] @ Rz)[perm])[None].float()
    cov1 = torch.cat([torch.rand(1, N, 3, generator=g) * 0.5 + 0.05, rn(1, N, 4)], -1).requires_grad_(True)
    cov2 = torch.cat([torch.rand(1, N, 3, generator=g) * 0.5 + 0.05, rn(1, N, 4)], -1).requires_grad_(True)
    qp = torch.tensor([[0.9999, 0.001, -0.002, 0.004]])
    qp = (qp / qp.norm()).requires_grad_(True)
    Tp = torch.tensor([[0.6, 0.02, 0.0]], requires_grad=True)
    import torchplus
    Rp = kornia.quaternion_to_rotation_matrix(torchplus.roll(qp, shift=-1, dim=-1))
    tgt_pts = (Rp[:, None] @ p2[..., None]).squeeze(-1) + Tp[:, None]
    n_tgt = (Rp[:, None].detach() @ n2[..., None]).squeeze(-1)
    closs = L.Aleat5_1ChamferL2NormalWeightedALLSVDLoss(loss_weight=1.0, penalize_ratio=0.97, norm=False,
                                                        pred_downsample_ratio=1, reg_weight=0.005, sph_weight=1)
    for icp_iter in (1, 2, 5):
        l, rr, tt = closs(p1, tgt_pts, cov_pred=cov1, cov_target=cov2, R_pred=Rp, t_pred=Tp,
                          normal_pred=n1.detach(), normal_target=n_tgt.detach(), mask=None, icp_iter=icp_iter)
        out["closs_l_%d" % icp_iter] = np_(l)
        out["closs_R_%d" % icp_iter] = np_(rr)
        out["closs_t_%d" % icp_iter] = np_(tt)
    l, rr, tt = closs(p1, tgt_pts, cov_pred=cov1, cov_target=cov2, R_pred=Rp, t_pred=Tp, normal_pred=n1.detach(),
                      normal_target=n_tgt.detach(), mask=None, icp_iter=2)
    l.backward()
    out.update(closs_p1=np_(p1), closs_n1=np_(n1), closs_p2=np_(p2), closs_n2=np_(n2), closs_cov1=np_(cov1),
               closs_cov2=np_(cov2), closs_q=np_(qp), closs_T=np_(Tp), closs_g_cov1=np_(cov1.grad),
               closs_g_cov2=np_(cov2.grad), closs_g_q=np_(qp.grad), closs_g_T=np_(Tp.grad))

    # ---------------------------------------------------------------- network: keys + reduced head + create_loss
    from google.protobuf import text_format
    from rslo.protos import pipeline_pb2
    from rslo.builder import second_builder, voxel_builder
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    with open(os.path.join(REF, "config/kitti_train_ours.prototxt")) as f:
        text_format.Merge(f.read(), cfg)
    mcfg = cfg.model.second

    # spconv stand-in modules so SpMiddleFHDWithCov2_3.__init__ can register parameters of the right shapes
    import spconv as sp

    class _SpConvStub(torch.nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     bias=True, indice_key=None):
            super().__init__()
            ks = [kernel_size] * 3 if isinstance(kernel_size, int) else list(kernel_size)
            self.weight = torch.nn.Parameter(torch.zeros(*ks, in_channels, out_channels))
            if bias:
                self.bias = torch.nn.Parameter(torch.zeros(out_channels))

    class _SpInvStub(_SpConvStub):
        def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True):
            super().__init__(in_channels, out_channels, kernel_size, bias=bias)

    sp.SparseConv3d = sp.SubMConv3d = _SpConvStub
    sp.SparseInverseConv3d = _SpInvStub
    sp.SparseSequential = torch.nn.Sequential
    vg = voxel_builder.build(mcfg.voxel_generator)
    torch.manual_seed(7)
    net = second_builder.build(mcfg, vg)
    sd = net.state_dict()
    keys = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump({"num_params": int(sum(p.numel() for p in net.parameters())), "keys": keys}, f, indent=0)
    grads_none = sorted(n for n, p in net.named_parameters() if not p.requires_grad)
    print("state dict:", len(keys), "frozen params:", grads_none)

    # reduced head: same class, small widths -> small state dict that fits a fixture
    import rslo.models.odom_pred as OP
    torch.manual_seed(11)
    head = OP.get_odom_class("UNRResNetOdomPredEncDecSVDTempMask")(
        bn_type="SyncBN", enc_use_norm=True, conv_type="mask_conv", layer_nums=[1, 2, 1], layer_strides=[2, 2, 2],
        num_filters=[8, 8, 16], upsample_strides=[2, 2, 2], num_upsample_filters=[8, 8, 8], num_input_features=12,
        pooling_type="avg_pool", pooling_size=1, dropout=1e-22, cycle_constraint=True, pred_pyramid_motion=True,
        use_deep_supervision=True, odom_format="rx+t", point_cloud_range=pc_range, dense_predict=True,
        conf_type="softmax", use_svd=False, cubic_pred_height=0)
    head.train()
    H, W = 16, 24
    xs = []
    for t in range(3):
        x = rn(2, 6, H, W)
        x = x * (torch.rand(2, 1, H, W, generator=g) > 0.6).float()
        xs.append(x)
    hsd = {k: np_(v).copy() for k, v in head.state_dict().items()}   # copy: numpy views alias the live buffers
    res = head([x.clone() for x in xs])
    after = {k: np_(v) for k, v in head.state_dict().items() if "running_" in k or "num_batches" in k}
    np.savez_compressed(os.path.join(HERE, "head_small.npz"),
                        **{"sd/" + k: v for k, v in hsd.items()},
                        **{"sd_after/" + k: v for k, v in after.items()},
                        **{"x%d" % i: np_(x) for i, x in enumerate(xs)},
                        t_pred=np_(res["translation_preds"][0]), r_pred=np_(res["rotation_preds"][0]),
                        tq_map_g=np_(res["tq_map_g"]), t_conf=np_(res["t_conf"]), r_conf=np_(res["r_conf"]),
                        **{"py%d_pred" % i: np_(p[0]) for i, p in enumerate(res["pyramid_motion"])},
                        **{"py%d_mask" % i: np_(p[1]) for i, p in enumerate(res["pyramid_motion"])})

    # create_loss assembly on the real network object (head outputs faked at 96x176 would be 40 MB: use the
    # reduced head's outputs and the full loss code path with step > 1500 and step <= 1500)
    net.odom_predictor = head
    net.train()
    for step in (2000, 100):
        net.global_step.fill_(step)
        feats, covs = [], []
        gg = torch.Generator().manual_seed(99)
        for t in range(3):
            n = 900 + 37 * t
            c = synthetic.small_cloud(n, seed=20 + t)
            feats.append(torch.from_numpy(c))
            covs.append(torch.cat([torch.rand(n, 3, generator=gg) * 0.5 + 0.05, torch.randn(n, 4, generator=gg)], -1))
        res = head([x.clone() for x in xs])
        res = {k: v for k, v in res.items()}
        # batch 2 in the head input but the reference loss is bs=1: use the first sample's 3 pairs
        sel = slice(0, 3)
        preds = {"translation_preds": [res["translation_preds"][0][sel]],
                 "rotation_preds": [res["rotation_preds"][0][sel]],
                 "pyramid_motion": [[p[0][sel], p[1][sel]] for p in res["pyramid_motion"]],
                 "middle_conf_preds": covs, "voxel_features": feats}
        example = {"icp_odometry": torch.zeros(3, 7), "tq_maps": [torch.zeros(3, 7, H, W)]}
        ret = net.loss(example, preds)
        tag = "cl%d_" % step
        out.update({tag + "loss": np_(ret["loss"]), tag + "T": np_(ret["translation_loss"]),
                    tag + "R": np_(ret["rotation_loss"]), tag + "py": np_(ret["pyramid_loss"]),
                    tag + "C": np_(ret["C_loss"]), tag + "tq_tgt": np_(example["tq_maps"][0])})
        if step == 2000:
            for t in range(3):
                out["cl_feat%d" % t] = np_(feats[t])
                out["cl_cov%d" % t] = np_(covs[t])

    np.savez_compressed(os.path.join(HERE, "ref_pieces.npz"), **out)
    for f in ("ref_pieces.npz", "head_small.npz", "state_dict_keys.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
