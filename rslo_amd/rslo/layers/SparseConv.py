"""(feature, mask)-pair aware normalisation / activation wrappers
(reference: rslo/layers/SparseConv.py:96-132,195-220): they act on the feature tensor and pass the
mask through untouched."""
import apex
import torch
import torch.nn as nn

from rslo.layers.normalization import MaskSyncBatchNorm, SemiGlobalSyncBatchNorm


def _pair(fn, x):
    if isinstance(x, (tuple, list)):
        return [fn(x[0]), x[1]]
    return fn(x)


class SPC_SyncBN2d(apex.parallel.SyncBatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None, channel_last=False, fuse_relu=False, noise_scale_std=0, noise_shift_std=0):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, process_group, channel_last)
        assert noise_scale_std == 0 and noise_shift_std == 0, "BN noise is not used by the RSLO hot path"
        self.add_noise = False

    def forward(self, x, act_slope=None, residual=None):
        """(feature, mask) pairs: the mask passes through; act_slope / residual are the optional fused epilogue of the
        ROCm SyncBatchNorm stand-in (y = act(bn(x) + residual))."""
        if isinstance(x, (tuple, list)):
            res = residual[0] if isinstance(residual, (tuple, list)) else residual
            return [self._feature(x[0], act_slope, res), x[1]]
        return self._feature(x, act_slope, residual)

    def _feature(self, x, act_slope, residual):
        if act_slope is None and residual is None:
            return super().forward(x)
        if hasattr(super(), "fusable"):
            return super().forward(x, act_slope=act_slope, residual=residual)
        y = super().forward(x)          # a real apex SyncBatchNorm: separate ops
        if residual is not None:
            y = y + residual
        if act_slope is not None:
            y = nn.functional.leaky_relu(y, act_slope) if act_slope != 0.0 else nn.functional.relu(y)
        return y


class SPC_MaskSyncBN2d(MaskSyncBatchNorm):
    """(reference: rslo/layers/SparseConv.py:22-58) statistics over the occupied cells; output re-masked."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None,
                 channel_last=False, fuse_relu=False, noise_scale_std=0, noise_shift_std=0):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, process_group, channel_last)
        assert noise_scale_std == 0 and noise_shift_std == 0, "BN noise is not used by the RSLO configurations"

    def forward(self, x):
        pair = isinstance(x, (tuple, list))
        tensor, mask = x if pair else (x, (x.abs().sum(dim=1, keepdim=True) > 0).float().detach())
        y = super().forward([tensor, mask]) * mask
        return [y, mask] if pair else y


class SPC_SemiGlobalSyncBN2d(SemiGlobalSyncBatchNorm):
    """(reference: rslo/layers/SparseConv.py:60-94)"""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None,
                 channel_last=False, fuse_relu=False, noise_scale_std=0, noise_shift_std=0):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, process_group, channel_last)
        assert noise_scale_std == 0 and noise_shift_std == 0, "BN noise is not used by the RSLO configurations"

    def forward(self, x):
        return _pair(super().forward, x)


class SPC_BN2d(nn.BatchNorm2d):
    def forward(self, x):
        return _pair(super().forward, x)


class SPC_ReLU(nn.ReLU):
    def forward(self, x):
        return _pair(super().forward, x)


class SPC_LeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        return _pair(super().forward, x)


def act_slope_of(m):
    """Negative slope of a (SPC_)ReLU / LeakyReLU module, None for anything else."""
    if isinstance(m, nn.LeakyReLU):
        return float(m.negative_slope)
    if isinstance(m, nn.ReLU):
        return 0.0
    return None


class SparseConv(nn.Module):
    """2-D mask-normalised convolution (reference: rslo/layers/SparseConv.py:222-302; conv_type "sparse_conv", not the
    shipped "mask_conv"): y = conv(x * m) / conv_ones(m) + b where the window holds any mask weight (0 elsewhere), and
    the mask is max-pooled (max_pool_mask) or sum-pooled and renormalised by the per-sample maximum of the INPUT mask.
    A (feature, mask) pair goes in and comes out; a bare tensor gets the mask `sum over channels != 0`."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, bias=True, max_pool_mask=True,
                 **kwargs):
        super().__init__()
        assert kernel_size % 2 == 1
        self.out_channels, self.use_bias, self.max_pool_mask = out_channels, bias, max_pool_mask
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, bias=False,
                               padding=padding)
        self.sum_conv = nn.Conv2d(1, 1, kernel_size, stride, bias=False, padding=padding)
        self.sum_conv.weight.requires_grad_(False)
        nn.init.constant_(self.sum_conv.weight, 1)
        if max_pool_mask:
            self.mask_pool = nn.MaxPool2d(kernel_size, stride=stride, padding=padding)
        else:
            self.mask_pool = nn.Conv2d(1, 1, kernel_size, stride, bias=False, padding=padding)
            self.mask_pool.weight.requires_grad_(False)
            nn.init.constant_(self.mask_pool.weight, 1)
        self.b = nn.ParameterList([nn.Parameter(torch.zeros(out_channels, 1, 1))]) if bias else [0]

    def sparse_conv(self, tensor, mask):
        if mask is None:
            mask = torch.ones_like(tensor[:, :1])
        mask = mask.detach()
        features = self.conv1(tensor * mask)
        norm = self.sum_conv(mask)
        norm = torch.where(norm == 0, torch.zeros_like(norm), 1.0 / (norm + 1e-12))
        feature = features * norm + self.b[0]
        if self.max_pool_mask:
            mask = self.mask_pool(mask)
        else:
            top = mask.reshape(mask.shape[0], -1).max(dim=-1)[0].view(-1, 1, 1, 1)
            mask = self.mask_pool(mask) / top
        return [feature, mask.detach()]

    def forward(self, x):
        if not isinstance(x, (list, tuple)):
            x = [x, (torch.sum(x, dim=1, keepdim=True) != 0).float()]
        return self.sparse_conv(x[0], x[1])


class _ConvBNActFn(torch.autograd.Function):
    """y = act(bn(conv3x3_s1(x) + bias)) as ONE autograd node (the skip / deblock / prediction trunks of the BEV head:
    Conv2d -> SyncBN -> ReLU triples outside the BasicBlocks).  Same kernels in the same order as the two nodes it
    replaces (rslo_conv2d_fwd, rslo_bn2d_*, rslo_conv2d_wgrad with the bias gradient from the same pass): bit-identical."""

    @staticmethod
    def forward(ctx, x, w, cb, g, b, conv, bn, slope, fork=False):
        from rslo_amd import capi, precision
        from apex.parallel import _world, count_batch, fused_bn_forward
        x_in = x
        x = x.contiguous()
        lp = precision.low_precision() is not None
        ws = getattr(w, "_hip_split", None)
        if ws is not None and (ws[2] != w._version or ws[3] != w.data_ptr()):
            ws = None
        wf, wt = (ws[0], ws[1]) if ws is not None else (capi.conv2d_wsplit(w, False), capi.conv2d_wsplit(w, True))
        group = bn.process_group
        world = _world(group)
        o = capi.conv2d_fwd(x, wf, cb, w.shape[0], lp=lp)
        if bn.track_running_stats:
            count_batch(bn)
        y, mean, invstd, cnt = fused_bn_forward(bn, o, None, g, b, slope, group, world)
        ctx.save_for_backward(x, w, g, o, y if slope != 1.0 else None, mean, invstd, cnt)
        ctx.meta = (wt, slope, lp, cb is not None, group, world)
        ctx.leaf_params = (w,) if cb is None else (w, cb)
        if fork:      # the input again, as an output of THIS node: what its other consumers read (see forward_fork)
            ctx.set_materialize_grads(False)
            return y, x_in
        return y

    @staticmethod
    def backward(ctx, gy, g_other=None):
        from rslo_amd import capi
        from rslo.layers import hip_conv2d
        from apex.parallel import fused_bn_backward
        x, w, g, o, y, mean, invstd, cnt = ctx.saved_tensors
        wt, slope, lp, has_bias, group, world = ctx.meta
        if gy is None:      # only the forked input was used downstream
            return g_other, None, None, None, None, None, None, None, None
        d_o, _, dg, db = fused_bn_backward(gy.contiguous(), y, o, g, mean, invstd, cnt, slope, False, True, group, world)
        dx = None
        if ctx.needs_input_grad[0]:      # the other consumers' gradient joins in the data gradient's epilogue
            dx = capi.conv2d_fwd(d_o, wt, None, w.shape[1], lp=lp,
                                 residual=None if g_other is None else g_other.contiguous())
        dcb = None
        if has_bias:
            dw, dcb = hip_conv2d.conv2d_wgrad_leaf(x, d_o, 1, want_bias=True, lp=lp, params=ctx.leaf_params)
        else:
            dw = hip_conv2d.conv2d_wgrad_leaf(x, d_o, 1, lp=lp, params=ctx.leaf_params)
        return dx, dw, dcb, dg, db, None, None, None, None


def _conv_bn_fusable(conv, bn, x):
    """Static + shape-dependent eligibility of a (Conv2d, SyncBN) pair for _ConvBNActFn, cached on the conv module."""
    import os
    from rslo.layers import hip_conv2d
    key = (tuple(x.shape), x.dtype, x.is_cuda, conv.training, bn.training, torch.is_grad_enabled())
    cached = conv.__dict__.get("_convbn_ok")
    if cached is not None and cached[0] == key:
        return cached[1]
    ok = (os.environ.get("RSLO_FUSED_BLOCK", "1") != "0" and isinstance(conv, hip_conv2d.Conv2d) and x.is_cuda
          and x.dtype == torch.float32 and x.dim() == 4 and bn.training and torch.is_grad_enabled()
          and hip_conv2d.HIP_PASSES == "wfd" and conv._kind() == "3x3" and conv.stride == (1, 1)
          and conv.weight.requires_grad and hasattr(bn, "fusable") and bn.fusable(x) and bn.affine
          and conv._eligible(x) and conv._hip_ok[1] and conv._hip_ok[2])
    conv.__dict__["_convbn_ok"] = (key, bool(ok))
    return bool(ok)


class FusedSequential(nn.Sequential):
    """nn.Sequential that hands a following (Leaky)ReLU to a normalisation layer able to fuse it (the ROCm
    SyncBatchNorm stand-in: one kernel pair per direction for BN + activation).  Same children, same state-dict keys,
    same results as nn.Sequential."""

    def _steps(self):
        """[(module, fused activation slope or None)], rebuilt after the children change."""
        steps = self.__dict__.get("_fused_steps")
        if steps is not None:
            return steps
        mods = list(self._modules.values())
        steps, i = [], 0
        from rslo.layers import hip_conv2d
        while i < len(mods):
            m = mods[i]
            slope = act_slope_of(mods[i + 1]) if i + 1 < len(mods) else None
            if (isinstance(m, hip_conv2d.Conv2d) and i + 2 < len(mods) and isinstance(mods[i + 1], SPC_SyncBN2d)
                    and act_slope_of(mods[i + 2]) is not None):
                steps.append(((m, mods[i + 1]), act_slope_of(mods[i + 2])))       # conv -> BN -> act: one node when eligible
                i += 3
            elif slope is not None and isinstance(m, SPC_SyncBN2d):
                steps.append((m, slope))
                i += 2
            else:
                steps.append((m, None))
                i += 1
        self.__dict__["_fused_steps"] = steps
        return steps

    def add_module(self, name, module):
        self.__dict__.pop("_fused_steps", None)
        super().add_module(name, module)

    def __setitem__(self, idx, module):
        self.__dict__.pop("_fused_steps", None)
        super().__setitem__(idx, module)

    def __delitem__(self, idx):
        self.__dict__.pop("_fused_steps", None)
        super().__delitem__(idx)

    def __setattr__(self, name, value):
        if isinstance(value, nn.Module) or name in self.__dict__.get("_modules", ()):
            self.__dict__.pop("_fused_steps", None)
        super().__setattr__(name, value)

    def forward(self, x):
        return self.forward_from(x, 0)

    def forward_fork(self, x):
        """-> (self(x), x'), x' holding x's values for the OTHER consumers of x.  When the first children are a fused
        conv -> BN -> activation node, x' is an output of that node and the gradient its consumers send back arrives as
        an operand of the node's backward, where the data-gradient kernel adds it in its epilogue -- instead of the
        autograd engine summing the two contributions to x with a launch of its own (seven full-map additions per step in
        the BEV head: every encoder stage feeds a skip branch, every decoder stage a prediction branch).  Otherwise
        x' is x."""
        steps = self._steps()
        if steps and isinstance(steps[0][0], tuple) and isinstance(x, torch.Tensor) and x.requires_grad:
            conv, bn = steps[0][0]
            if _conv_bn_fusable(conv, bn, x):
                y, x_other = _ConvBNActFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, conv, bn, steps[0][1], True)
                return self._run(y, steps[1:], 0), x_other
        return self.forward_from(x, 0), x

    def forward_from(self, x, first_child):
        """The children from index `first_child` on (a caller that has already applied the leading ones, e.g. the
        head's fused concatenation + nn.Upsample in front of a deblock)."""
        return self._run(x, self._steps(), first_child)

    @staticmethod
    def _run(x, steps, first_child):
        skip = first_child
        for m, slope in steps:
            if skip > 0:      # steps are whole children or (conv, BN, act) / (BN, act) groups: only leading singles are skipped
                if isinstance(m, tuple) or slope is not None:
                    raise ValueError("forward_from: child %d is inside a fused group" % first_child)
                skip -= 1
                continue
            if isinstance(m, tuple):
                conv, bn = m
                if isinstance(x, torch.Tensor) and _conv_bn_fusable(conv, bn, x):
                    x = _ConvBNActFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, conv, bn, slope)
                else:
                    x = bn(conv(x), act_slope=slope)
            else:
                x = m(x) if slope is None else m(x, act_slope=slope)
        return x
