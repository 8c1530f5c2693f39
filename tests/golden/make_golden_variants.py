"""Golden vectors for registry variants outside the shipped configuration, from the reference's own
code imported from /root/reference:  python tests/golden/make_golden_variants.py  ->  tests/golden/variants_ref.npz

  * `aggregate_tq` with use_svd=True (rslo/models/odom_pred.py:315-346, rslo/utils/geometric.py:159-218,
    rslo/layers/svd.py): scene-flow rigid fit over the occupied BEV cells, per sample;
  * the 2-D `SparseConv` layer (rslo/layers/SparseConv.py:222-302): mask-normalised convolution + mask pooling, both
    mask modes, forward and gradients;
  * MaskSyncBatchNorm / SemiGlobalSyncBatchNorm (rslo/layers/normalization.py:11-251), two training steps + one eval
    call each.  They hand their statistics to apex's SyncBatchnormFunction, which is neither in this image nor in the
    reference tree: a restatement of its published Python fallback is injected for the run (`ApexSyncBNFn` below), so
    these vectors pin the REFERENCE'S statistics logic (mask rectifier, drift-adapted momenta, running buffers), not apex.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref as MG   # noqa: E402


class ApexSyncBNFn(torch.autograd.Function):
    """apex/parallel/sync_batchnorm_kernel.py (Python fallback), single process."""

    @staticmethod
    def forward(ctx, x, weight, bias, mean, var, eps, group, world):
        ctx.save_for_backward(x, weight, mean, var)
        ctx.eps = eps
        sh = [1, -1, 1, 1]
        return (x - mean.view(sh)) / torch.sqrt(var.view(sh) + eps) * weight.view(sh) + bias.view(sh)

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, var = ctx.saved_tensors
        sh = [1, -1, 1, 1]
        xmu = x - mean.view(sh)
        mean_dy = gy.mean((0, 2, 3))
        mean_dy_xmu = (gy * xmu).mean((0, 2, 3))
        gx = (gy - mean_dy.view(sh) - xmu / (var.view(sh) + ctx.eps) * mean_dy_xmu.view(sh)) / torch.sqrt(var.view(sh) + ctx.eps)
        gx = gx * weight.view(sh)
        gw = (gy * xmu / torch.sqrt(var.view(sh) + ctx.eps)).sum((0, 2, 3))
        return gx, gw, gy.sum((0, 2, 3)), None, None, None, None, None


def norm_variants(out, g):
    sys.modules["apex.parallel.sync_batchnorm_kernel"].SyncBatchnormFunction = ApexSyncBNFn
    torch.cuda.nvtx.range_push = lambda *a, **k: None
    torch.cuda.nvtx.range_pop = lambda *a, **k: None
    import rslo.layers.normalization as RN
    RN.SyncBatchnormFunction = ApexSyncBNFn      # the name the module bound at import time
    for name, cls in (("mask", RN.MaskSyncBatchNorm), ("semi", RN.SemiGlobalSyncBatchNorm)):
        bn = cls(5, eps=1e-3, momentum=0.01)
        bn.process_group = None          # attribute of the real apex class that the import shim lacks
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, 5))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, 5))
        bn.train()
        for step in range(2):
            x = (torch.randn(2, 5, 6, 7, generator=g) * 1.5 + 0.3).requires_grad_(True)
            mk = (torch.rand(2, 1, 6, 7, generator=g) > 0.4).float()
            y = bn([x * 1.0, mk]) if name == "mask" else bn(x * 1.0)
            (y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
            tag = "%s%d_" % (name, step)
            out.update({tag + "x": MG.np_(x), tag + "m": MG.np_(mk), tag + "y": MG.np_(y), tag + "gx": MG.np_(x.grad),
                        tag + "gw": MG.np_(bn.weight.grad), tag + "gb": MG.np_(bn.bias.grad)})
            bn.zero_grad()
            for k, v in bn.state_dict().items():
                if k not in ("weight", "bias"):
                    out[tag + "sd/" + k] = MG.np_(v).copy()      # copy: += updates num_batches_tracked in place
        bn.eval()
        x = torch.randn(2, 5, 6, 7, generator=g)
        mk = (torch.rand(2, 1, 6, 7, generator=g) > 0.4).float()
        y = bn([x.clone(), mk]) if name == "mask" else bn(x)
        out.update({name + "E_x": MG.np_(x), name + "E_m": MG.np_(mk), name + "E_y": MG.np_(y)})


def main():
    MG.install_shims()
    import rslo.models.odom_pred as OP
    from rslo.layers.SparseConv import SparseConv
    g = torch.Generator().manual_seed(77)
    out = {}
    pc_range = np.array([-70.4, -38.4, -3, 70.4, 38.4, 5], np.float32)
    B, H, W = 3, 12, 22
    tq = torch.randn(B, 7, H, W, generator=g) * 0.3
    tq[:, 3] += 2.0
    tq[:, 3:] = tq[:, 3:] / tq[:, 3:].norm(dim=1, keepdim=True)
    mask = torch.rand(B, 1, H, W, generator=g) > 0.4
    t_conf = torch.rand(B, 1, H, W, generator=g) * mask
    r_conf = torch.rand(B, 1, H, W, generator=g) * mask
    fake = types.SimpleNamespace(use_svd=True, point_cloud_range=pc_range)
    odom = OP.UNOdomPredEncDecSVDTempMaskBase.aggregate_tq(fake, [tq], [mask], [t_conf], [r_conf])[0]
    out.update(svd_tq=MG.np_(tq), svd_mask=MG.np_(mask), svd_tconf=MG.np_(t_conf), svd_rconf=MG.np_(r_conf),
               svd_odom=MG.np_(odom))

    for mode in (True, False):
        torch.manual_seed(5)
        m = SparseConv(6, 8, kernel_size=3, stride=2, padding=1, bias=True, max_pool_mask=mode)
        with torch.no_grad():
            m.b[0].normal_(0, 0.1)
        x = (torch.randn(2, 6, 10, 14, generator=g)).requires_grad_(True)
        mk = (torch.rand(2, 1, 10, 14, generator=g) > 0.5).float() * (1 + torch.rand(2, 1, 10, 14, generator=g))
        y, mo = m([x, mk])
        (y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
        tag = "spc%d_" % int(mode)
        out.update({tag + "x": MG.np_(x), tag + "mask": MG.np_(mk), tag + "w": MG.np_(m.conv1.weight),
                    tag + "b": MG.np_(m.b[0]), tag + "y": MG.np_(y), tag + "mask_out": MG.np_(mo),
                    tag + "gx": MG.np_(x.grad), tag + "gw": MG.np_(m.conv1.weight.grad), tag + "gb": MG.np_(m.b[0].grad)})
    norm_variants(out, g)
    path = os.path.join(HERE, "variants_ref.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
