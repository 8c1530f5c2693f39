"""The sparse encoder and the VFE against vectors produced by the REFERENCE'S OWN forward code
(tests/golden/make_golden_encoder.py: rslo/models/middle.py:219-245 and voxel_encoder.py:258-280 run over a dense
conv3d / conv_transpose3d emulation of the spconv operator on a 41 x 64 x 64 crop).

  * not gpu : the host mirror over the CPU oracle (oracle/cpu_backend.py) -- pins the oracle's sparse conv,
              rulebook chain and the mirror's layer graph to the reference's graph
  * gpu     : the same mirror on the HIP kernels through the C ABI

Tolerances (fp32, 20 chained layers, |activations| up to ~150): outputs and the input gradient 2e-5 of the largest
entry, parameter gradients 1e-4 (dense emulation sums in a different order; LeakyReLU kinks are measure-zero on
these inputs), BatchNorm running statistics 5e-6, level-2 site list bit-exact.
"""
import os
import sys

import numpy as np
import pytest
import torch

import rslo_amd  # noqa: F401
from oracle import cpu_backend

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from golden_weights import fill_parameters  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "encoder_ref.npz")
SEED = 31


def rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def lin(shape, seed):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(tuple(shape)).astype(np.float32))


def build_encoder(bn_type):
    from rslo.models.middle import SpMiddleFHDWithCov2_3
    enc = SpMiddleFHDWithCov2_3(output_shape=[1, 40, 64, 64, 16], bn_type=bn_type, use_leakyReLU=True,
                                num_input_features=7)
    fill_parameters(enc, SEED)
    return enc.train()


def check_encoder(g, device, bn_type, tag):
    enc = build_encoder(bn_type).to(device)
    feats = torch.from_numpy(g["vfe_out"]).to(device).requires_grad_(True)
    c = g["coords"]
    coors = torch.from_numpy(np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)).to(device)
    lvl2 = {}
    enc.middle_conv.register_forward_hook(lambda m, i, o: lvl2.update(f=o.features.detach(), idx=o.indices))
    bev, cov = enc(feats, coors, 1)
    ((bev * lin(bev.shape, 5).to(device)).sum() + (cov * lin(cov.shape, 6).to(device)).sum()).backward()
    assert bev.shape == (1, 128, 8, 8)
    assert rel(bev, g[tag + "bev"]) < 2e-5
    assert rel(cov, g[tag + "cov"]) < 2e-5
    # strided-conv output sites: same set in the same (ascending linear index) order as the dense emulation
    assert int(g[tag + "lvl2_n"]) == lvl2["f"].shape[0]
    assert (lvl2["idx"].cpu().numpy() == g[tag + "lvl2_idx"]).all()
    assert rel(lvl2["f"].double().sum(0), g[tag + "lvl2_sum"]) < 2e-5
    assert rel(lvl2["f"].double().abs().sum(0), g[tag + "lvl2_abs"]) < 2e-5
    assert rel(feats.grad, g[tag + "g_feats"]) < 2e-5
    sd = enc.state_dict()
    n_bn = 0
    for k in g.files:
        if k.startswith(tag + "sd_after/"):
            name = k[len(tag) + 9:]
            if "num_batches" in name:
                assert int(sd[name]) == int(g[k])
            else:
                assert rel(sd[name], g[k]) < 5e-6, name
            n_bn += 1
    assert n_bn == (15 if bn_type == "None" else 57)
    checked = 0
    # a bias followed by BatchNorm has an analytically zero gradient (rounding noise on both sides)
    before_bn = {n for n, _ in enc.named_parameters() if n.endswith(".bias") and
                 n[:-5].rsplit(".", 1)[0] + "." + str(int(n[:-5].rsplit(".", 1)[1]) + 1) + ".running_mean" in sd}
    for name, p in enc.named_parameters():
        if name in before_bn:
            assert float(p.grad.abs().max()) < 1e-3
            continue
        if tag + "g/" + name in g.files:
            assert rel(p.grad, g[tag + "g/" + name]) < 1e-4, name
        else:
            gk, gc = p.grad.sum(dim=(3, 4)).reshape(-1), p.grad.sum(dim=(0, 1, 2))
            assert rel(gk, g[tag + "gk/" + name]) < 1e-4, name
            assert rel(gc, g[tag + "gc/" + name]) < 1e-4, name
        checked += 1
    assert checked + len(before_bn) == len(list(enc.parameters()))
    assert len(before_bn) == (5 if bn_type == "None" else 19)


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def test_oracle_voxelizer_reproduces_fixture_inputs(g):
    """The committed crop is what the oracle voxelizer makes of the synthetic scan (fixture and code in step)."""
    import oracle as O
    from rslo_amd import synthetic
    p0, _, _ = synthetic.frame_pair(0)
    v, c, n = O.voxelize(p0, [4.0, -3.2, -3.0, 10.4, 3.2, 5.0], [0.1, 0.1, 0.2], 10, 20000)
    assert (c == g["coords"]).all() and (n == g["num_points"]).all() and (v == g["voxels"]).all()


def test_vfe_oracle_matches_reference(g):
    import oracle as O
    out = O.vfe_mean(g["voxels"], g["num_points"])
    assert rel(out, g["vfe_out"]) < 1e-6


@pytest.mark.parametrize("bn_type,tag", [("None", ""), ("BN", "bn_")])
def test_encoder_over_oracle_matches_reference(g, bn_type, tag):
    with cpu_backend.patched():
        check_encoder(g, "cpu", bn_type, tag)


@pytest.mark.gpu
def test_vfe_gpu_matches_reference(g, hip):
    from rslo.models.voxel_encoder import SimpleVoxel_XYZINormalC
    vfe = SimpleVoxel_XYZINormalC(num_input_features=7)
    out = vfe(torch.from_numpy(g["voxels"]).cuda(), torch.from_numpy(g["num_points"]).cuda(),
              torch.from_numpy(g["coords"]).cuda())
    assert rel(out, g["vfe_out"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("bn_type,tag", [("None", ""), ("BN", "bn_")])
def test_encoder_gpu_matches_reference(g, hip, bn_type, tag):
    check_encoder(g, "cuda", bn_type, tag)
