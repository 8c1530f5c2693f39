"""CPU stand-in for rslo_amd.capi built on the oracle -- TEST INFRASTRUCTURE ONLY.

`with cpu_backend.patched():` swaps the entry points of `rslo_amd.capi` for oracle-backed versions that
accept CPU tensors, so the *same* host modules (spconv mirror, encoder, loss) can be run on the CPU as the
reference-semantics path: the checker for the GPU parity tests and the "port" CPU baseline of bench.py.
The product never imports this module and has no switch that could route to it.
"""
import contextlib

import numpy as np
import torch

import oracle as O


def _np(t):
    return t.detach().cpu().numpy()


class SiteIndex:
    def __init__(self, coords, batch, dims):
        self.coords = coords.contiguous()
        self.batch = int(batch)
        self.dims = [int(d) for d in dims]
        self.cap = 0
        self.keys = self.vals = None


def voxelize(points, pc_range, voxel_size, grid_xyz, max_points, max_voxels):
    v, c, n = O.voxelize(_np(points), pc_range, voxel_size, max_points, max_voxels)
    M = len(c)
    F = points.shape[1]
    vox = torch.zeros((max_voxels, max_points, F))
    coords = torch.zeros((max_voxels, 3), dtype=torch.int32)
    num = torch.zeros((max_voxels,), dtype=torch.int32)
    vox[:M], coords[:M], num[:M] = torch.from_numpy(v), torch.from_numpy(c), torch.from_numpy(n)
    return vox, coords, num, torch.tensor([M], dtype=torch.int32)


def _is64(*ts):
    return any(isinstance(t, torch.Tensor) and t.dtype == torch.float64 for t in ts)


def _conv64(x, W, nbr, transpose):
    """float64 arbiter of the sparse conv on a neighbour table: y[o] = sum_k x[nbr[o,k]] @ W[k] (or W[k]^T) -- used
    when the host modules are run in double precision to separate rounding from errors (tests/test_gpu_model.py)."""
    n, K = nbr.shape
    W = W.reshape(K, W.shape[-2], W.shape[-1])
    y = np.zeros((n, W.shape[1] if transpose else W.shape[2]))
    for k in range(K):
        rows = np.nonzero(nbr[:, k] >= 0)[0]
        if len(rows):
            y[rows] += x[nbr[rows, k]] @ (W[k].T if transpose else W[k])
    return y


def vfe_mean(voxels, num_points):
    if _is64(voxels):
        m = voxels.sum(1) / num_points.to(voxels.dtype).reshape(-1, 1)
        return torch.cat([m[:, :4], m[:, 4:7] / (m[:, 4:7].norm(dim=-1, keepdim=True) + 1e-12)], 1)
    return torch.from_numpy(O.vfe_mean(_np(voxels), _np(num_points)))


def rulebook_subm(index, ks):
    return torch.from_numpy(O.rulebook_subm(_np(index.coords), index.batch, index.dims, ks))


def conv_out_dims(in_dims, ks, stride, pad):
    return O.conv_out_dims(in_dims, ks, stride, pad)


def rulebook_conv(index, ks, stride, pad):
    oc, od, nbr, nbrT = O.rulebook_conv(_np(index.coords), index.batch, index.dims, ks, stride, pad)
    return SiteIndex(torch.from_numpy(oc), index.batch, od), torch.from_numpy(nbr), torch.from_numpy(nbrT)


def spconv_fwd(x, W, bias, nbr, flip_k=False, act_slope=1.0, order=None):
    Wn = _np(W)
    if flip_k:
        Wn = Wn[::-1].copy()
    if _is64(x, W):
        y = _conv64(_np(x).astype(np.float64), Wn.astype(np.float64), _np(nbr), False)
        if bias is not None:
            y = y + _np(bias).astype(np.float64)
        if act_slope != 1.0:
            y = np.where(y > 0, y, y * act_slope)
        return torch.from_numpy(y)
    y = O.spconv_fwd(_np(x), Wn, None if bias is None else _np(bias), _np(nbr))
    if act_slope != 1.0:
        y = np.where(y > 0, y, y * np.float32(act_slope)).astype(np.float32)
    return torch.from_numpy(y)


def spconv_dgrad(dout, W, nbrT, flip_k=False, order=None):
    Wn = _np(W)
    if flip_k:
        Wn = Wn[::-1].copy()
    if _is64(dout, W):
        return torch.from_numpy(_conv64(_np(dout).astype(np.float64), Wn.astype(np.float64), _np(nbrT), True))
    return torch.from_numpy(O.spconv_dgrad(_np(dout), Wn, _np(nbrT)))


def spconv_wgrad(x, dout, nbr, cin, cout, with_bias=True):
    if _is64(x, dout):
        t, xd, gd = _np(nbr), _np(x).astype(np.float64), _np(dout).astype(np.float64)
        dW = np.zeros((t.shape[1], cin, cout))
        for k in range(t.shape[1]):
            rows = np.nonzero(t[:, k] >= 0)[0]
            if len(rows):
                dW[k] = xd[t[rows, k]].T @ gd[rows]
        return torch.from_numpy(dW), (torch.from_numpy(gd.sum(0)) if with_bias else None)
    dW, db = O.spconv_wgrad(_np(x), _np(dout), _np(nbr), cin, cout)
    return torch.from_numpy(dW), (torch.from_numpy(db) if with_bias else None)


def rulebook_row_order(nbr, flip_k=False):
    return None         # a scheduling hint of the GPU kernels; results do not depend on it


def rulebook_pairs(nbr):
    t = _np(nbr)
    n, K = t.shape
    pin, pout, koff = [], [], [0]
    for k in range(K):
        rows = np.nonzero(t[:, k] >= 0)[0]
        pin.append(t[rows, k])
        pout.append(rows.astype(np.int32))
        koff.append(koff[-1] + len(rows))
    return (torch.from_numpy(np.concatenate(pin).astype(np.int32)), torch.from_numpy(np.concatenate(pout).astype(np.int32)),
            torch.tensor(koff, dtype=torch.int32))


def spconv_wgrad_pairs(x, dout, pairs, n_out, K, cin, cout, with_bias=True, bias_partial=None):
    pin, pout, koff = (_np(p) for p in pairs)
    xd, gd = _np(x).astype(np.float64), _np(dout).astype(np.float64)
    dW = np.zeros((K, cin, cout), np.float64)
    for k in range(K):
        a, b = int(koff[k]), int(koff[k + 1])
        if b > a:
            dW[k] = xd[pin[a:b]].T @ gd[pout[a:b]]
    if _is64(x, dout):
        return torch.from_numpy(dW), (torch.from_numpy(gd.sum(0)) if with_bias else None)
    db = torch.from_numpy(gd.sum(0).astype(np.float32)) if with_bias else None
    return torch.from_numpy(dW.astype(np.float32)), db


def leaky_bwd(y, dout, slope, colsum=False):
    g = torch.where(y > 0, dout, dout * slope)
    return (g, None) if colsum else g


def dense_scatter(feat, coords, batch, dims):
    if result_dtype == torch.float64 and feat.dtype == torch.float32:
        feat = feat.double()      # C4 arbiter: the bf16 trunk leaves through `.float()`; the double head takes it from here
    if _is64(feat):
        out = torch.zeros(batch, *[int(d) for d in dims], feat.shape[1], dtype=feat.dtype)
        c = coords.long()
        out[c[:, 0], c[:, 1], c[:, 2], c[:, 3]] = feat
        return out.permute(0, 4, 1, 2, 3).contiguous()
    return torch.from_numpy(O.dense(_np(feat), _np(coords), batch, list(dims)))


def dense_gather(dense, coords, C_, batch, dims):
    c = coords.long()
    return dense[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].contiguous()


def chamfer_nn(xyz1, xyz2, dist=None, idx=None, ncnt=None, mcnt=None):
    if _is64(xyz1, xyz2):       # neighbour choice by the fp32 search, distance recomputed in double
        _, i = chamfer_nn(xyz1.float(), xyz2.float(), None, None, ncnt, mcnt)
        nb = torch.gather(xyz2, 1, i.long()[..., None].expand(-1, -1, 3))
        d = ((xyz1 - nb) ** 2).sum(-1)
        if ncnt is not None:
            d = torch.where(torch.arange(d.shape[1])[None] < torch.as_tensor(ncnt).reshape(-1, 1).long(), d,
                            torch.full_like(d, float("inf")))
        return d, i
    if ncnt is None and mcnt is None:
        d, i = O.chamfer_nn(_np(xyz1), _np(xyz2))
    else:       # ragged batch: pair by pair on the valid prefixes, padding rows get +inf / 0
        a, b = _np(xyz1), _np(xyz2)
        B, N, _ = a.shape
        d = np.full((B, N), np.inf, np.float32)
        i = np.zeros((B, N), np.int32)
        for k in range(B):
            n = int(ncnt[k]) if ncnt is not None else N
            m = int(mcnt[k]) if mcnt is not None else b.shape[1]
            dk, ik = O.chamfer_nn(a[k:k + 1, :n], b[k:k + 1, :m])
            d[k, :n], i[k, :n] = dk[0], ik[0]
    d, i = torch.from_numpy(d), torch.from_numpy(i)
    if dist is not None:
        dist.copy_(d)
        idx.copy_(i)
        return dist, idx
    return d, i


def chamfer_grad(xyz1, xyz2, graddist1, idx1, g1=None, g2=None):
    if _is64(xyz1, xyz2, graddist1):
        nb = torch.gather(xyz2, 1, idx1.long()[..., None].expand(-1, -1, 3))
        a = 2 * graddist1[..., None] * (xyz1 - nb)
        b = torch.zeros_like(xyz2).scatter_add_(1, idx1.long()[..., None].expand(-1, -1, 3), -a)
        return a, b
    a, b = O.chamfer_grad(_np(xyz1), _np(xyz2), _np(graddist1), _np(idx1))
    a, b = torch.from_numpy(a), torch.from_numpy(b)
    if g1 is not None:
        g1.copy_(a)
        g2.copy_(b)
        return g1, g2
    return a, b


# ---------------------------------------------------------------------------------------------------------------
# C4 (bf16 operands): the same rounding points as the HIP path, accumulation in float64.
#   sparse trunk (rslo_spconv_fwd_bf16 / rslo_leaky_bwd_colsum_bf16 / rslo_spconv_wgrad_pairs_bf16): rows ARE bfloat16
#   tensors, master weights rounded to bf16 per call, bias / accumulation wide, outputs rounded to bf16;
#   dense 3x3 stride-1 layers (rslo_conv2d_fwd_bf16 / rslo_conv2d_wgrad_bf16): activations, output gradients and
#   weights rounded to bf16 as operands, accumulation and storage wide.
# `result_dtype` = dtype of the master weights of the network being run (float32 path or float64 arbiter): the dtype
# weight / bias gradients are returned in.
# ---------------------------------------------------------------------------------------------------------------
result_dtype = torch.float32


def _bf(t):
    """value rounded to bfloat16 (round to nearest even, like v_cvt_pk_bf16_f32), kept in t's dtype"""
    return t.float().to(torch.bfloat16).to(t.dtype)


def spconv_fwd_bf16(x, W, bias, nbr, flip_k=False, act_slope=1.0, transpose=False, order=None):
    Wn = _np(_bf(W.float()).double())
    if flip_k:
        Wn = Wn[::-1].copy()
    y = _conv64(_np(x.double()), Wn, _np(nbr), transpose)
    if bias is not None:
        y = y + _np(bias.double())
    if act_slope != 1.0:
        y = np.where(y > 0, y, y * act_slope)
    return torch.from_numpy(y).float().to(torch.bfloat16)       # fp32 accumulator -> bf16 row, as the kernel stores it


def leaky_bwd_bf16(y, dout, slope, colsum=False):
    g = torch.where(y.float() > 0, dout.float(), dout.float() * slope).to(torch.bfloat16)
    return (g, g.double().sum(0, keepdim=True).float()) if colsum else g


def spconv_wgrad_pairs_bf16(x, dout, pairs, n_out, K, cin, cout, bias_partial=None):
    pin, pout, koff = (_np(p) for p in pairs)
    xd, gd = _np(x.double()), _np(dout.double())
    dW = np.zeros((K, cin, cout), np.float64)
    for k in range(K):
        a, b = int(koff[k]), int(koff[k + 1])
        if b > a:
            dW[k] = xd[pin[a:b]].T @ gd[pout[a:b]]
    db = None if bias_partial is None else bias_partial.double().sum(0).to(result_dtype)
    return torch.from_numpy(dW).to(result_dtype), db


class _LpConv3x3Fn(torch.autograd.Function):
    """3x3 / stride-1 / padding-1 convolution with bf16-rounded operands in all three passes (what rslo_conv2d_fwd_bf16
    and rslo_conv2d_wgrad_bf16 compute), accumulated in float64."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        y = torch.nn.functional.conv2d(_bf(x).double(), _bf(w).double(), None if bias is None else bias.double(), 1, 1)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        g = _bf(dy).double()
        dx = torch.nn.grad.conv2d_input(x.shape, _bf(w).double(), g, 1, 1).to(x.dtype)
        dw = torch.nn.grad.conv2d_weight(_bf(x).double(), w.shape, g, 1, 1).to(w.dtype)
        db = g.sum((0, 2, 3)).to(w.dtype) if ctx.has_bias else None     # the kernel sums its (rounded) dout operands
        return dx, dw, db


def _lp_conv_forward(orig):
    def fwd(self, input, weight, bias):
        from rslo_amd import capi, precision
        if (precision.low_precision() is not None and not input.is_cuda and input.dim() == 4
                and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1
                and capi.conv2d_fwd_supported(self.in_channels, self.out_channels, input.shape[2], input.shape[3])
                and capi.conv2d_wgrad_supported(self.in_channels, self.out_channels, input.shape[2], input.shape[3], 1)):
            return _LpConv3x3Fn.apply(input, weight, bias)
        return orig(self, input, weight, bias)
    return fwd


_NAMES = ["spconv_fwd_bf16", "leaky_bwd_bf16", "spconv_wgrad_pairs_bf16", "SiteIndex", "voxelize", "vfe_mean", "rulebook_subm", "conv_out_dims", "rulebook_conv", "spconv_fwd",
          "spconv_dgrad", "spconv_wgrad", "rulebook_row_order", "rulebook_pairs", "spconv_wgrad_pairs", "leaky_bwd", "dense_scatter", "dense_gather", "chamfer_nn", "chamfer_grad"]


@contextlib.contextmanager
def patched():
    import spconv
    from rslo.layers import hip_conv2d
    from rslo_amd import capi, precision
    saved = {n: getattr(capi, n) for n in _NAMES}
    lowp, convf = spconv.SparseConvolution._low_precision, hip_conv2d.Conv2d._conv_forward
    try:
        for n in _NAMES:
            setattr(capi, n, globals()[n])
        # C4 on CPU tensors: the product gates its bf16 paths on `is_cuda`; the emulation takes the same layers
        spconv.SparseConvolution._low_precision = lambda self, feats: (
            precision.low_precision() is not None and self.allow_low_precision
            and self.in_channels in (32, 64) and self.out_channels in (32, 64))
        hip_conv2d.Conv2d._conv_forward = _lp_conv_forward(convf)
        yield
    finally:
        for n, f in saved.items():
            setattr(capi, n, f)
        spconv.SparseConvolution._low_precision, hip_conv2d.Conv2d._conv_forward = lowp, convf
