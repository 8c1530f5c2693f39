"""BasicBlock of the BEV encoder (reference: rslo/models/custom_resnet_spc.py:224-298): two 3x3
(mask-)convs with BN, residual add where masks are averaged, ReLU."""
import torch
import torch.nn as nn

from rslo.layers.SparseConv import SPC_LeakyReLU, SPC_ReLU, SPC_SyncBN2d, act_slope_of


def conv1x1(in_planes, out_planes, stride=1, Conv2d=None, groups=1):
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=0, bias=False, groups=groups)


def conv3x3(in_planes, out_planes, stride=1, Conv2d=None, groups=1):
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False, groups=groups)


def _mask_mean(ma, mb):
    if ma is None or mb is None:        # mask declared unused by the caller (MaskConv passes None through)
        return None
    return ((ma + mb) / 2).float()


def SPC_add(a, b):
    if isinstance(a, (list, tuple)):
        return [a[0] + b[0], _mask_mean(a[1], b[1])]
    return a + b


def SPC_cat(a, b):
    if isinstance(a, (list, tuple)):
        return [torch.cat([a[0], b[0]], dim=1), _mask_mean(a[1], b[1])]
    return torch.cat([a, b], dim=1)


class c2_probe:
    """Shape stand-in for conv2's input (conv1's output) so that Conv2d._eligible can be asked before it exists."""

    def __init__(self, x, stride):
        self.shape = (x.shape[0], x.shape[1], (x.shape[2] - 1) // stride + 1, (x.shape[3] - 1) // stride + 1)


def _split_of(w, ksize3):
    """(forward operand, data-gradient operand) of a conv weight: from the one-launch presplit when still valid."""
    from rslo_amd import capi
    ws = getattr(w, "_hip_split", None)
    if ws is not None and ws[2] == w._version and ws[3] == w.data_ptr():
        return ws[0], ws[1]
    if ksize3:
        return capi.conv2d_wsplit(w, False), capi.conv2d_wsplit(w, True)
    return capi.conv2d_wsplit_k(w, False), capi.conv2d_wsplit_k(w, True)


class _BasicBlockFn(torch.autograd.Function):
    """One autograd node for a whole BasicBlock:  y = act(bn2(conv2(act(bn1(conv1(x))))) + shortcut(x)),
    shortcut = identity or bn_d(conv1x1_s2(x)).  The same kernels in the same order as the layer-by-layer path
    (rslo_conv2d_fwd / _fwd_s2, rslo_bn2d_fwd_local, their backward counterparts) -- results are bit-identical -- but 1
    node instead of 4-6: the interpreter / autograd-engine time per block drops by ~70 us per step, and the shortcut
    gradient is added to the data gradient right here instead of by the engine's accumulation."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, blk, hip_w1):
        from rslo_amd import capi, precision
        from apex.parallel import _world, count_batch, fused_bn_forward
        x = x.contiguous()
        lp = precision.low_precision() is not None
        slope = act_slope_of(blk.relu)
        s = blk.stride
        planes = w1.shape[0]

        group = blk.bn1.process_group
        world = _world(group)

        def bn_fwd(bn, t, res, g, b, sl):
            if bn.track_running_stats:
                count_batch(bn)
            return fused_bn_forward(bn, t, res, g, b, sl, group, world)
        w1f, w1t = _split_of(w1, True)
        w2f, w2t = _split_of(w2, True)
        o1 = capi.conv2d_fwd_s2(x, w1f, planes, 3) if s == 2 else capi.conv2d_fwd(x, w1f, None, planes, lp=lp)
        y1, m1, i1, n1 = bn_fwd(blk.bn1, o1, None, g1, b1, slope)
        o2 = capi.conv2d_fwd(y1, w2f, None, planes, lp=lp)
        wdt = od = md = idd = nd = None
        if wd is not None:
            wdf, wdt = _split_of(wd, False)
            od = capi.conv2d_fwd_s2(x, wdf, planes, 1)
            res, md, idd, nd = bn_fwd(blk.downsample[1], od, None, gd, bd, 1.0)
        else:
            res = x
        y2, m2, i2, n2 = bn_fwd(blk.bn2, o2, res, g2, b2, slope)
        ctx.save_for_backward(x, w1, g1, w2, g2, wd, gd, o1, y1, m1, i1, n1, o2, y2, m2, i2, n2, od, md, idd, nd)
        ctx.ops = (w1t, w2t, wdt)
        ctx.meta = (slope, s, lp, hip_w1, group, world)
        return y2

    @staticmethod
    def backward(ctx, gy):
        from rslo_amd import capi
        from apex.parallel import fused_bn_backward
        x, w1, g1, w2, g2, wd, gd, o1, y1, m1, i1, n1, o2, y2, m2, i2, n2, od, md, idd, nd = ctx.saved_tensors
        w1t, w2t, wdt = ctx.ops
        slope, s, lp, hip_w1, group, world = ctx.meta
        gy = gy.contiguous()
        act = slope != 1.0
        d_o2, d_res, dg2, db2 = fused_bn_backward(gy, y2 if act else None, o2, g2, m2, i2, n2, slope, True, True, group, world)
        d_y1 = capi.conv2d_fwd(d_o2, w2t, None, w2.shape[1], lp=lp)
        from rslo.layers import hip_conv2d
        dw2 = hip_conv2d.conv2d_wgrad_leaf(y1, d_o2, 1, lp=lp, params=(w2,))
        d_o1, _, dg1, db1 = fused_bn_backward(d_y1, y1 if act else None, o1, g1, m1, i1, n1, slope, False, True, group, world)
        res_joined = False
        if s == 2:
            dx = capi.conv2d_dgrad_s2(d_o1, w1t, w1.shape[1], x.shape[2], x.shape[3], 3)
        else:       # identity block: the residual branch's gradient joins in the epilogue (same bits as dx.add_(d_res))
            dx = capi.conv2d_fwd(d_o1, w1t, None, w1.shape[1], lp=lp, residual=d_res if wd is None else None)
            res_joined = wd is None
        if hip_w1:
            dw1 = hip_conv2d.conv2d_wgrad_leaf(x, d_o1, s, lp=lp, params=(w1,))
        else:       # the full-resolution stride-2 layer: the library's weight gradient (see csrc/conv2d.hip conv2d_plan)
            from rslo_amd import streams      # leaf work, like the hand-written weight gradients (rslo_amd/streams.py)
            dw1 = streams.leaf(lambda: torch.ops.aten.convolution_backward(
                d_o1, x, w1, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1], (d_o1, x), (w1,))
        dwd = dgd = dbd = None
        if wd is not None:
            from rslo_amd import streams
            d_od, _, dgd, dbd = fused_bn_backward(d_res, None, od, gd, md, idd, nd, 1.0, False, True, group, world)
            # the downsample branch's input gradient lands on every other pixel of every other row: added INTO the 3x3
            # branch's gradient (this node's own buffer), a quarter of the map touched instead of a full read + write
            dx = capi.conv2d_dgrad_s2(d_od, wdt, wd.shape[1], x.shape[2], x.shape[3], 1, residual=dx, inplace=True)
            dwd = hip_conv2d.conv1x1s2_wgrad_leaf(x, d_od, wd.shape, params=(wd,))
        elif not res_joined:
            dx.add_(d_res)
        return dx, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None, None


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, lrelu=False, BN=None, Conv2d=None, groups=1,
                 use_se=False, use_sa=False):
        super().__init__()
        assert not use_se and not use_sa, "SE / spatial attention are off in the shipped config"
        self.BatchNorm2d = BN if BN is not None else nn.BatchNorm2d
        self.Conv2d = Conv2d if Conv2d is not None else nn.Conv2d
        self.conv1 = conv3x3(inplanes, planes, stride, Conv2d=self.Conv2d, groups=groups)
        self.bn1 = self.BatchNorm2d(planes)
        self.relu = SPC_LeakyReLU(0.1, True) if lrelu else SPC_ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes, Conv2d=self.Conv2d, groups=groups)
        self.bn2 = self.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.use_se = self.use_sa = False

    def _fused_node_ok(self, x):
        """The whole block as one autograd node (_BasicBlockFn): training, every layer on the hand-written kernels (any
        number of ranks: the SyncBN statistics exchange happens inside the node, as in _FusedBNActFn).
        RSLO_FUSED_BLOCK=0 keeps the layer-by-layer nodes."""
        key = (tuple(x.shape), x.dtype, x.is_cuda, self.training, torch.is_grad_enabled())
        cached = self.__dict__.get("_fused_ok")
        if cached is None or cached[0] != key:
            cached = self.__dict__["_fused_ok"] = (key, self._fused_node_check(x))
        return cached[1]

    def _fused_node_check(self, x):
        import os
        from rslo.layers import hip_conv2d
        if os.environ.get("RSLO_FUSED_BLOCK", "1") == "0" or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            return False
        if not (self.training and torch.is_grad_enabled() and hip_conv2d.HIP_PASSES == "wfd"):
            return False
        c1, c2 = getattr(self.conv1, "conv1", self.conv1), getattr(self.conv2, "conv1", self.conv2)
        bns = [self.bn1, self.bn2]
        convs = [c1, c2]
        if self.downsample is not None:
            mods = list(self.downsample.children()) if isinstance(self.downsample, nn.Sequential) else []
            if len(mods) != 2:
                return False
            cd = getattr(mods[0], "conv1", mods[0])
            if not (isinstance(cd, hip_conv2d.Conv2d) and cd._kind() == "1x1s2" and cd.stride == (2, 2)):
                return False
            bns.append(mods[1])
            convs.append(cd)
        for bn in bns:
            if not (isinstance(bn, SPC_SyncBN2d) and hasattr(bn, "fusable") and bn.fusable(x) and bn.affine
                    and bn.process_group is bns[0].process_group):
                return False
        for c in convs[:2]:
            if not (isinstance(c, hip_conv2d.Conv2d) and c._kind() == "3x3" and c.bias is None and c.weight.requires_grad):
                return False
        if self.stride not in (1, 2) or c1.stride != (self.stride, self.stride) or c2.stride != (1, 1):
            return False
        if (self.downsample is None) != (self.stride == 1 and c1.in_channels == c1.out_channels):
            return False
        if not c1._eligible(x) or not c1._hip_ok[2]:
            return False
        if not (c2._eligible(c2_probe(x, self.stride)) and c2._hip_ok[1] and c2._hip_ok[2]):
            return False
        self._fused_hip_w1 = bool(c1._hip_ok[1])
        return True

    def forward(self, x):
        pair = isinstance(x, (list, tuple))
        if (not pair or x[1] is None) and self._fused_node_ok(x[0] if pair else x):
            t = x[0] if pair else x
            c1, c2 = getattr(self.conv1, "conv1", self.conv1), getattr(self.conv2, "conv1", self.conv2)
            if self.downsample is not None:
                ds = list(self.downsample.children())
                cd = getattr(ds[0], "conv1", ds[0])
                wd, gd, bd = cd.weight, ds[1].weight, ds[1].bias
            else:
                wd = gd = bd = None
            y = _BasicBlockFn.apply(t, c1.weight, self.bn1.weight, self.bn1.bias, c2.weight, self.bn2.weight,
                                    self.bn2.bias, wd, gd, bd, self, self._fused_hip_w1)
            return [y, None] if pair else y
        if isinstance(self.bn1, SPC_SyncBN2d) and isinstance(self.bn2, SPC_SyncBN2d):
            # BN + ReLU and BN + residual add + ReLU as fused epilogues of the normalisation kernels
            slope = act_slope_of(self.relu)
            out = self.bn1(self.conv1(x), act_slope=slope)
            out = self.conv2(out)
            residual = x if self.downsample is None else self.downsample(x)
            y = self.bn2(out, act_slope=slope, residual=residual)
            if isinstance(y, (list, tuple)):      # (feature, mask) pairs: masks are averaged like SPC_add does
                y = [y[0], _mask_mean(out[1], residual[1])]
            return y
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(SPC_add(out, residual))
