"""spconv operator API used by the RSLO hot path, implemented on librslo_hip.so.

Mirrors the subset of spconv-1.x that rslo/models/middle.py:80-97,119-213,224,240 touches:
SparseConvTensor, SubMConv3d, SparseConv3d, SparseInverseConv3d, SparseSequential (+ utils.VoxelGenerator).
Same constructor arguments, parameter names/shapes (weight [kz,ky,kx,Cin,Cout], bias [Cout]) and
indice_key sharing rules, so reference checkpoints load and the reference modules run unchanged.

MI355X design (not spconv's gather -> GEMM -> scatter-add per offset):
  * a level's active sites are indexed by a device hash (capi.SiteIndex);
  * a rulebook is a dense neighbour table [rows, K] (int32, -1 = no neighbour) plus its transpose;
    strided-conv outputs are numbered by ascending linear index via a bitmap rank (no sort);
  * convolution is ONE output-stationary implicit-GEMM launch per layer (MFMA fp32), with bias and
    a following LeakyReLU/ReLU fused into the epilogue by SparseSequential;
  * SubM tables are cached on the site index, so `dsubm*` keys on a level that already ran `subm*`
    reuse the table instead of rebuilding it.
There is no CPU path: tensors must be on the GPU and the HIP library must be present.
"""
import math

import torch
from torch import nn

from rslo_amd import capi

from . import utils  # noqa: F401

__all__ = ["SparseConvTensor", "SparseModule", "SparseSequential", "SubMConv3d", "SparseConv3d",
           "SparseInverseConv3d", "utils"]


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


class Rulebook:
    """Neighbour tables of one indice_key."""

    def __init__(self, kind, nbr, nbrT, in_index, out_index, ks, stride, pad):
        self.kind = kind            # "subm" | "conv"
        self.nbr = nbr              # [n_out, K]: input row per (out, k)
        self.nbrT = nbrT            # [n_in, K]:  output row per (in, k)   (None for subm: symmetric)
        self.in_index = in_index
        self.out_index = out_index
        self.ks, self.stride, self.pad = ks, stride, pad
        self._pairs = None
        self._orders = {}

    def order(self, which):
        """Scheduling order of a table's rows for the tile-granular kernels (capi.rulebook_row_order), built once.
        Only the transposed table of a strided convolution ("nbrT": inverse-conv forward, strided data gradient) gets
        one: its rows have 8 of 27 neighbours and mask-sorted tiles run 25-30 % faster.  The dense SubM / forward tables
        ("nbr", "nbr_flip") stay in raster order -- there the scattered gathers cost what the skipped offsets save
        (measured, csrc/spconv.hip)."""
        if which != "nbrT" or self.nbrT is None:
            return None
        if which not in self._orders:
            self._orders[which] = capi.rulebook_row_order(self.nbrT)
        return self._orders[which]

    def pairs(self):
        """(pairs_in, pairs_out, koff): the table as spconv-style pair lists, built once per rulebook."""
        if self._pairs is None:
            self._pairs = capi.rulebook_pairs(self.nbr)
        return self._pairs

    def indice_pairs(self):
        """spconv-1.x view of the table: (indice_pairs [K,2,P] padded with -1, indice_num [K]);
        pairs of offset k in ascending input row.  For parity checks / introspection only."""
        t = self.nbrT if self.nbrT is not None else torch.flip(self.nbr, dims=[1])
        N, K = t.shape
        num = (t >= 0).sum(0).to(torch.int32)
        cap = max(int(num.max().item()) if N else 0, 1)
        pairs = torch.full((K, 2, cap), -1, dtype=torch.int32, device=t.device)
        for k in range(K):
            rows = torch.nonzero(t[:, k] >= 0).flatten()
            pairs[k, 0, : rows.numel()] = rows.to(torch.int32)
            pairs[k, 1, : rows.numel()] = t[rows, k]
        return pairs, num


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, index=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        self._index = index

    @property
    def spatial_size(self):
        return int(torch.tensor(self.spatial_shape).prod())

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def site_index(self):
        if self._index is None:
            idx = self.indices
            if idx.dtype != torch.int32:
                idx = idx.int()
            self._index = capi.SiteIndex(idx.contiguous(), self.batch_size, self.spatial_shape)
            self._index.subm_cache = {}
        return self._index

    def batch_offsets(self):
        """Row range of every batch element (rows are grouped by batch index on every level:
        level 0 is a concatenation, strided outputs are in ascending linear order).  One host read,
        cached on the site index."""
        index = self.site_index()
        offs = getattr(index, "batch_offs", None)
        if offs is None:
            b = index.coords[:, 0].contiguous()
            probe = torch.arange(self.batch_size + 1, dtype=b.dtype, device=b.device)
            dev = torch.searchsorted(b, probe).to(torch.int32)
            index.batch_offs_dev = dev
            offs = dev.tolist()
            index.batch_offs = offs
        return offs

    def dense(self, channels_first=True, frames=1, out=None):
        """frames > 1 (not part of spconv's API): the batch holds frame t of sample b at index t * (batch / frames) + b and
        the result is [batch / frames, frames, C, D, H, W] -- the frames of a sample side by side (rslo_dense_scatter_frames).
        out (not part of spconv's API): a buffer to write the dense tensor into (capi.dense_scatter)."""
        feats = self.features.float() if self.features.dtype == torch.bfloat16 else self.features   # bf16 trunk (C4)
        # (the buffer travels in a list: it is a destination, not an autograd input)
        out = _DenseFn.apply(feats, self.site_index().coords, self.batch_size, tuple(self.spatial_shape), int(frames),
                             None if out is None else [out])
        if not channels_first:
            assert frames == 1
            out = out.permute(0, 2, 3, 4, 1).contiguous()
        return out

    def _like(self, features, indices=None, spatial_shape=None, index=None):
        t = SparseConvTensor(features, self.indices if indices is None else indices,
                             self.spatial_shape if spatial_shape is None else spatial_shape, self.batch_size,
                             self.grid, index if index is not None else (self._index if indices is None else None))
        t.indice_dict = self.indice_dict
        return t


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, coords, batch, dims, frames=1, out=None):
        ctx.save_for_backward(coords)
        ctx.meta = (feat.shape[1], batch, dims, frames)
        kw = {} if out is None else {"out": out[0]}
        if frames == 1:
            return capi.dense_scatter(feat.contiguous(), coords, batch, dims, **kw)
        return capi.dense_scatter(feat.contiguous(), coords, batch, dims, frames, **kw)

    @staticmethod
    def backward(ctx, g):
        (coords,) = ctx.saved_tensors
        C_, batch, dims, frames = ctx.meta
        if frames == 1:
            return capi.dense_gather(g.contiguous(), coords, C_, batch, dims), None, None, None, None, None
        return capi.dense_gather(g.contiguous(), coords, C_, batch, dims, frames), None, None, None, None, None


def _w3(weight, K, cin, cout):
    """[kd,kh,kw,cin,cout] parameter -> [K,cin,cout] view that carries the parameter's pre-split operands (presplit())."""
    W3 = weight.reshape(K, cin, cout)
    pre = getattr(weight, "_hip_split", None)
    if pre is not None:
        W3._hip_split = pre
    return W3


def presplit(root):
    """Refresh the split-bf16 operands of every 32/64-channel sparse convolution under `root` (forward and data-gradient
    orientation) in ONE launch; capi.weight_split() then reuses them while the parameters are unchanged."""
    if not capi.SPLIT_BF16:
        return
    ent = root.__dict__.get("_hip_spconv_plan")
    if ent is None or any(w.data_ptr() != p for w, p in zip(ent[1], ent[0]["ptrs"])):
        ws = [m.weight for m in root.modules()
              if isinstance(m, SparseConvolution) and m.weight.is_cuda and m.weight.dtype == torch.float32
              and m.in_channels in (32, 64) and m.out_channels in (32, 64)]
        if not ws:
            return
        plan, views = capi.weight_split_many([w.reshape(-1, w.shape[-2], w.shape[-1]) for w in ws])
        ent = (plan, ws, views)
        root.__dict__["_hip_spconv_plan"] = ent
    capi.weight_split_run(ent[0])
    for w, (f, t) in zip(ent[1], ent[2]):
        w._hip_split = (f, t, w._version, w.data_ptr())


class _SparseConvFn(torch.autograd.Function):
    """y = act(bias + sum_k x[nbr[:,k]] @ W[k]) and its dgrad/wgrad, all through the C ABI."""

    @staticmethod
    def forward(ctx, x, weight, bias, nbr, nbrT, subm, slope, rb=None, inverse=False):
        ctx.rb, ctx.inverse = rb, inverse
        K = nbr.shape[1]
        cin, cout = weight.shape[-2], weight.shape[-1]
        W3 = _w3(weight, K, cin, cout)
        x = x.contiguous()
        # table walked forward / backward: (nbr, nbrT) of the rulebook, swapped for an inverse conv
        fwd_key, bwd_key = ("nbrT", "nbr") if inverse else ("nbr", "nbr_flip" if subm else "nbrT")
        ctx.bwd_key = bwd_key
        live = getattr(rb, "n_live", None) if rb is not None else None      # capacity-laid-out plan (rslo_amd/inference.py)
        if live is not None and live.get(fwd_key) is not None and not torch.is_grad_enabled():
            y = capi.spconv_fwd(x, W3, bias, nbr, flip_k=False, act_slope=slope, order=rb.order(fwd_key),
                                n_live=live[fwd_key])
        else:
            y = capi.spconv_fwd(x, W3, bias, nbr, flip_k=False, act_slope=slope,
                                order=None if rb is None else rb.order(fwd_key))
        ctx.save_for_backward(x, weight, y if slope != 1.0 else None, nbr, nbrT)
        ctx.meta = (subm, slope, bias is not None)
        ctx.params = (weight,) if bias is None else (weight, bias)      # (the objects whose .grad the results become)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y, nbr, nbrT = ctx.saved_tensors
        subm, slope, has_bias = ctx.meta
        K = nbr.shape[1]
        cin, cout = weight.shape[-2], weight.shape[-1]
        W3 = _w3(weight, K, cin, cout)
        g = gy.contiguous()
        bias_partial = None
        if slope != 1.0:
            want = has_bias and ctx.needs_input_grad[2] and ctx.rb is not None
            g = capi.leaky_bwd(y, g, slope, colsum=want)
            if want:
                g, bias_partial = g
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if subm:      # pair (i -> o via k)  <=>  (o -> i via K-1-k)
                gx = capi.spconv_dgrad(g, W3, nbr, flip_k=True, order=None if ctx.rb is None else ctx.rb.order(ctx.bwd_key))
            else:
                gx = capi.spconv_dgrad(g, W3, nbrT, flip_k=False, order=None if ctx.rb is None else ctx.rb.order(ctx.bwd_key))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            if ctx.rb is not None:
                pin, pout, koff = ctx.rb.pairs()
                # an inverse conv runs over the same pairs with the roles of the two sides swapped
                pairs = (pout, pin, koff) if ctx.inverse else (pin, pout, koff)
                # the chunk partials -> gradient stage is left to ONE launch at the end of the pass where nothing can read the
                # gradient earlier (rslo_amd/streams.py deferred_reduce)
                from rslo_amd import streams
                defer = streams.deferred_reduce(ctx.params) if all(isinstance(p, nn.Parameter) for p in ctx.params) else None
                if defer is not None:
                    with defer:
                        gw, gb = capi.spconv_wgrad_pairs(x, g, pairs, g.shape[0], K, cin, cout, with_bias=has_bias,
                                                         bias_partial=bias_partial)
                else:
                    gw, gb = capi.spconv_wgrad_pairs(x, g, pairs, g.shape[0], K, cin, cout, with_bias=has_bias,
                                                     bias_partial=bias_partial)
            else:
                gw, gb = capi.spconv_wgrad(x, g, nbr, cin, cout, with_bias=has_bias)
            gw = gw.reshape(weight.shape)
        return gx, gw, gb, None, None, None, None, None, None


class _SparseConvBf16Fn(torch.autograd.Function):
    """C4 (bf16 features, int32 rulebook, fp32 accumulate): rows in / out are bfloat16, weights and bias fp32 masters
    rounded to bf16 operands per call; forward = rslo_spconv_fwd_bf16, data gradient = the same kernel on the transposed
    operand, activation backward + bias partials = rslo_leaky_bwd_colsum_bf16, weight gradient =
    rslo_spconv_wgrad_pairs_bf16 (fp32 dW / dbias).  Channel counts 32 / 64 on both sides."""

    @staticmethod
    def forward(ctx, x, weight, bias, nbr, nbrT, subm, slope, rb, inverse):
        ctx.rb, ctx.inverse = rb, inverse
        K = nbr.shape[1]
        cin, cout = weight.shape[-2], weight.shape[-1]
        W3 = weight.reshape(K, cin, cout)
        x = x.contiguous()
        fwd_key, bwd_key = ("nbrT", "nbr") if inverse else ("nbr", "nbr_flip" if subm else "nbrT")
        ctx.bwd_key = bwd_key
        y = capi.spconv_fwd_bf16(x, W3, bias, nbr, act_slope=slope, order=rb.order(fwd_key))
        ctx.save_for_backward(x, weight, y if slope != 1.0 else None, nbr, nbrT)
        ctx.meta = (subm, slope, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y, nbr, nbrT = ctx.saved_tensors
        subm, slope, has_bias = ctx.meta
        K = nbr.shape[1]
        cin, cout = weight.shape[-2], weight.shape[-1]
        W3 = weight.reshape(K, cin, cout)
        g = gy.contiguous()
        want_b = has_bias and ctx.needs_input_grad[2]
        part = None
        if slope != 1.0:
            g = capi.leaky_bwd_bf16(y, g, slope, colsum=want_b)
            if want_b:
                g, part = g
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            table, flip = (nbr, True) if subm else (nbrT, False)
            gx = capi.spconv_fwd_bf16(g, W3, None, table, flip_k=flip, transpose=True, order=ctx.rb.order(ctx.bwd_key))
        if ctx.needs_input_grad[1] or want_b:
            pin, pout, koff = ctx.rb.pairs()
            pairs = (pout, pin, koff) if ctx.inverse else (pin, pout, koff)
            gw, gb = capi.spconv_wgrad_pairs_bf16(x, g, pairs, g.shape[0], K, cin, cout, bias_partial=part)
            if want_b and gb is None:         # layer without activation: column sums of the bf16 gradient in fp32
                gb = g.float().sum(0)
            gw = gw.reshape(weight.shape)
        return gx, gw, gb, None, None, None, None, None, None


class _SegBNActFn(torch.autograd.Function):
    """Per-frame BatchNorm1d (training statistics) + LeakyReLU through rslo_segbn_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, seg_off, S, max_len, slope):
        x = x.contiguous()
        y, mean, invstd = capi.segbn_fwd(x, seg_off, S, max_len, gamma, beta, bn.running_mean, bn.running_var,
                                         bn.momentum, bn.eps, slope)
        ctx.save_for_backward(x, y, gamma, mean, invstd, seg_off)
        ctx.meta = (S, max_len, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, mean, invstd, seg_off = ctx.saved_tensors
        S, max_len, slope = ctx.meta
        gx, dgamma, dbeta = capi.segbn_bwd(x, y, gy.contiguous(), seg_off, S, max_len, gamma, mean, invstd, slope)
        return gx, dgamma, dbeta, None, None, None, None, None


def _pending_batch_counts():
    """The open defer_batch_counts() list of the ROCm apex stand-in (compat/apex/parallel.py), or None."""
    import sys
    mod = sys.modules.get("apex.parallel")
    return getattr(mod, "_pending_counts", None) if mod is not None else None


def _segmented_bn_act(x, bn, slope):
    """Training-mode nn.BatchNorm1d over each frame of the batched tensor, fused with the activation."""
    index = x.site_index()
    offs = x.batch_offsets()
    dev_off = index.batch_offs_dev
    S = x.batch_size
    max_len = max(offs[b + 1] - offs[b] for b in range(S))
    feats = x.features.float() if x.features.dtype == torch.bfloat16 else x.features
    y = _SegBNActFn.apply(feats, bn.weight, bn.bias, bn, dev_off, S, max_len, slope)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        n_seen = sum(1 for b in range(S) if offs[b + 1] > offs[b])     # one training forward per non-empty frame
        pending = _pending_batch_counts()
        if pending is not None:       # inside defer_batch_counts(): one multi-tensor add for the whole network forward
            pending.extend([bn.num_batches_tracked] * n_seen)
        else:
            bn.num_batches_tracked.add_(n_seen)
    return x._like(y)


class SparseModule(nn.Module):
    """Marker base class: modules that consume / produce SparseConvTensor."""


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, output_padding=0, transposed=False, inverse=False,
                 indice_key=None, fused_bn=False):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not fused_bn
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.dilation = _triple(dilation)
        assert self.dilation == [1, 1, 1], "dilation is not used by the RSLO hot path"
        self.subm, self.inverse = subm, inverse
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%s, padding=%s, subm=%s, inverse=%s, indice_key=%s" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.subm,
            self.inverse, self.indice_key)

    # -- rulebook --------------------------------------------------------------------------
    def _rulebook(self, x):
        rb = x.find_indice_pair(self.indice_key)
        if self.inverse:
            assert rb is not None and rb.kind == "conv", "inverse conv needs the rulebook of its forward twin"
            return rb
        if rb is not None:
            return rb
        index = x.site_index()
        if self.subm:
            key = tuple(self.kernel_size)
            rb = index.subm_cache.get(key)
            if rb is None:
                nbr = capi.rulebook_subm(index, self.kernel_size)
                # no back-reference to `index`: the cache entry would close a reference cycle (index -> cache -> rulebook
                # -> index) and every batch's tables (~200 MB) would then wait for the cyclic garbage collector instead of
                # being freed when the example goes out of scope -- measured: live memory 3.6 -> 8.9 GB over 100 steps and
                # two hipMalloc calls per step; a SubM layer never reads these fields
                rb = Rulebook("subm", nbr, None, None, None, self.kernel_size, [1, 1, 1], None)
                index.subm_cache[key] = rb
        else:
            out_index, nbr, nbrT = capi.rulebook_conv(index, self.kernel_size, self.stride, self.padding)
            out_index.subm_cache = {}
            rb = Rulebook("conv", nbr, nbrT, index, out_index, self.kernel_size, self.stride, self.padding)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        return rb

    def plan(self, x):
        """Build (or look up) this layer's rulebook and return the output site set WITHOUT touching features.
        Rulebooks depend only on coordinates; planning a whole encoder up front puts the few host reads of
        strided-conv output counts before any convolution is enqueued, so the conv launches that follow run
        back to back on the GPU."""
        rb = self._rulebook(x)
        if self.inverse:
            return x._like(None, rb.in_index.coords, rb.in_index.dims, rb.in_index)
        if self.subm:
            return x._like(None)
        return x._like(None, rb.out_index.coords, rb.out_index.dims, rb.out_index)

    allow_low_precision = True      # False keeps a layer fp32 under amp O1 (the covariance branch)

    def _low_precision(self, feats):
        """C4: the 32/64-channel layers take bf16 rows when apex.amp O1 is on (rslo_amd.precision)."""
        from rslo_amd import precision
        return (precision.low_precision() is not None and self.allow_low_precision and feats.is_cuda
                and self.in_channels in (32, 64) and self.out_channels in (32, 64))

    def forward(self, x, act_slope=1.0):
        assert isinstance(x, SparseConvTensor)
        rb = self._rulebook(x)
        feats = x.features
        if self._low_precision(feats):
            fn = _SparseConvBf16Fn
            if feats.dtype != torch.bfloat16:
                feats = feats.to(torch.bfloat16)      # entering the bf16 trunk (differentiable cast)
        else:
            fn = _SparseConvFn
            if feats.dtype != self.weight.dtype:
                feats = feats.to(self.weight.dtype)   # leaving it (fp32 layers: 16-channel levels, covariance branch)
        if self.inverse:      # output sites = the saved INPUT sites of the forward twin, same order
            y = fn.apply(feats, self.weight, self.bias, rb.nbrT, rb.nbr, False, act_slope, rb, True)
            return x._like(y, rb.in_index.coords, rb.in_index.dims, rb.in_index)
        if self.subm:
            y = fn.apply(feats, self.weight, self.bias, rb.nbr, None, True, act_slope, rb, False)
            return x._like(y)
        y = fn.apply(feats, self.weight, self.bias, rb.nbr, rb.nbrT, False, act_slope, rb, False)
        return x._like(y, rb.out_index.coords, rb.out_index.dims, rb.out_index)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         subm=True, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)


def _act_slope(m):
    """Slope of an activation SparseSequential can fuse into the preceding conv, else None."""
    if type(m) is nn.LeakyReLU:
        return float(m.negative_slope)
    if type(m) is nn.ReLU:
        return 0.0
    return None


def _is_identity(m):
    return type(m).__name__ == "Empty" or isinstance(m, nn.Identity)


class SparseSequential(SparseModule):
    """nn.Sequential over sparse and dense modules: dense modules see `.features`
    (middle.py:181 mixes raw nn.BatchNorm1d into the chain).  conv -> [identity] -> (Leaky)ReLU
    runs as ONE fused launch."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for name, m in kwargs.items():
            self.add_module(name, m)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(str(len(self._modules)) if name is None else name, module)

    def plan(self, x):
        """Propagate only the site sets through the chain (see SparseConvolution.plan)."""
        for m in self._modules.values():
            if isinstance(m, (SparseConvolution, SparseSequential)):
                x = m.plan(x)
            elif isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training and x.indices.is_cuda:
                x.batch_offsets()      # per-frame statistics need the row ranges: read them here, not mid-forward
        return x

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseConvolution):
                j = i + 1
                while j < len(mods) and _is_identity(mods[j]):
                    j += 1
                slope = _act_slope(mods[j]) if j < len(mods) else None
                if slope is not None:
                    x = m(x, act_slope=slope)
                    i = j + 1
                    continue
                x = m(x)
            elif isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                fusable_bn = (type(m) is nn.BatchNorm1d and m.training and m.affine and m.track_running_stats
                              and m.momentum is not None and x.features.is_cuda and x.features.shape[0] > 0)
                if fusable_bn:
                    j = i + 1
                    slope = _act_slope(mods[j]) if j < len(mods) else None
                    x = _segmented_bn_act(x, m, 1.0 if slope is None else slope)
                    i = j + 1 if slope is not None else j
                    continue
                if (type(m) is nn.BatchNorm1d and not m.training and m.track_running_stats and x.features.is_cuda
                        and x.features.dtype == torch.float32 and x.features.shape[0] > 0 and not torch.is_grad_enabled()):
                    # eval / no-grad (evaluate.py): running statistics + the activation that follows in ONE launch
                    j = i + 1
                    slope = _act_slope(mods[j]) if j < len(mods) else None
                    live = getattr(x.site_index(), "n_live", None) if x._index is not None else None
                    x = x._like(capi.bn1d_eval_act(x.features.contiguous(), m.running_mean, m.running_var, m.weight, m.bias,
                                                   m.eps, 1.0 if slope is None else slope, n_live=live))
                    i = j + 1 if slope is not None else j
                    continue
                if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training and x.batch_size > 1:
                    # the reference runs the encoder one frame at a time (middle.py:221), so batch
                    # statistics are per frame: normalise each batch segment on its own, in frame order
                    offs = x.batch_offsets()
                    parts = [m(x.features[offs[b]:offs[b + 1]]) for b in range(x.batch_size)
                             if offs[b + 1] > offs[b]]
                    x = x._like(torch.cat(parts, 0))
                elif x.features.shape[0] > 0 or not isinstance(m, nn.modules.batchnorm._BatchNorm):
                    x = x._like(m(x.features))
            else:
                x = m(x)
            i += 1
        return x
