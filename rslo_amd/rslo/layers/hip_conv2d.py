"""Conv2d of the BEV head on the hand-written gfx950 kernels (csrc/conv2d.hip).

The reference builds its BEV encoder-decoder from torch.nn.Conv2d (rslo/models/odom_pred.py:65-134,398-426,
rslo/layers/MaskConv.py:20-73) and leaves all three convolution passes to cuDNN.  Here the weight gradient of every
3x3 / padding-1 layer whose shape the kernel takes goes through rslo_conv2d_wgrad (csrc/conv2d.hip): one kernel + one
reduce instead of MIOpen's split-K igemm + 4 layout transposes + a zero fill, fixed summation order.  Forward and data
gradient of the stride-1 layers go through rslo_conv2d_fwd with operands split once per step for the whole head
(presplit); stride-2 and 1x1 layers, eval / no-grad calls and CPU tensors stay plain nn.Conv2d.  Same parameters, same
state-dict keys."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# which passes of an eligible layer run on the hand-written kernels: "w" weight gradient, "f" forward, "d" data gradient
# ("" = everything on the library, for A/B measurements)
HIP_PASSES = os.environ.get("RSLO_CONV2D_PASSES", "wfd")
# "0": the stride-2 layers (3x3 and the 1x1 downsample) stay on the library (A/B measurements)
HIP_STRIDE2 = os.environ.get("RSLO_CONV2D_S2", "1") != "0"


def join_leaf_stream(device=None):
    from rslo_amd import streams
    streams.join(device)


def conv2d_wgrad_leaf(x, dy, stride, want_bias=False, lp=False, params=None):
    """rslo_conv2d_wgrad as leaf work of the backward pass (rslo_amd.streams); params = the parameters the results are
    the gradients of (weight[, bias]): without them the call cannot be checked and runs on the issuing stream."""
    from rslo_amd import capi, streams
    if want_bias:
        return streams.leaf(lambda: capi.conv2d_wgrad(x, dy, stride, want_bias=True, lp=lp), (x, dy), params)
    return streams.leaf(lambda: capi.conv2d_wgrad(x, dy, stride, lp=lp), (x, dy), params)


def conv1x1s2_wgrad_leaf(x, dy, wshape, params=None):
    """Weight gradient of a 1x1 / stride-2 downsample conv as leaf work: rslo_conv1x1s2_wgrad (the centre tap of the
    stride-2 kernel, fixed summation order), or a batched GEMM over the sampled pixels for shapes outside its range."""
    from rslo_amd import capi, streams

    def run():
        x_, dy_ = x.contiguous(), dy.contiguous()
        if x_.is_cuda and x_.dtype == torch.float32 and dy_.dtype == torch.float32:
            dw = capi.conv1x1s2_wgrad(x_, dy_)
            if dw is not None:
                return dw.reshape(wshape)
        xs = x_[:, :, ::2, ::2].flatten(2)                              # [B, cin, P]
        return torch.matmul(dy_.flatten(2), xs.transpose(1, 2)).sum(0).reshape(wshape)
    return streams.leaf(run, (x, dy), params)


def _low_precision():
    from rslo_amd import precision
    return precision.low_precision() is not None


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, hip_fd, hip_w):
        x = x.contiguous()
        ctx.hip_w = hip_w
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        ctx.has_bias = bias is not None
        ctx.leaf_params = (w,) if bias is None else (w, bias)
        ctx.hip_fd = hip_fd
        ctx.ws_t = None
        from rslo_amd import precision
        ctx.lp = lp = precision.low_precision() is not None      # C4: bf16 operands, fp32 accumulate / storage
        if hip_fd:
            from rslo_amd import capi
            ws = getattr(w, "_hip_split", None)      # operands refreshed for all layers in one launch (presplit())
            if ws is not None and (ws[2] != w._version or ws[3] != w.data_ptr()):
                ws = None
            if "d" in HIP_PASSES:
                ctx.ws_t = ws[1] if ws is not None else capi.conv2d_wsplit(w, True)
            if "f" in HIP_PASSES:
                wf = ws[0] if ws is not None else capi.conv2d_wsplit(w, False)
                if stride == 2:
                    return capi.conv2d_fwd_s2(x, wf, w.shape[0], 3)
                return capi.conv2d_fwd(x, wf, bias, w.shape[0], lp=lp)
        return F.conv2d(x, w, bias, stride, 1)

    @staticmethod
    def backward(ctx, dy):
        from rslo_amd import capi
        x, w = ctx.saved_tensors
        s = ctx.stride
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.ws_t is not None and s == 2:
                dx = capi.conv2d_dgrad_s2(dy, ctx.ws_t, w.shape[1], x.shape[2], x.shape[3], 3)
            elif ctx.ws_t is not None:
                dx = capi.conv2d_fwd(dy, ctx.ws_t, None, w.shape[1], lp=ctx.lp)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if ctx.hip_w and want_db and s == 1:
                dw, db = conv2d_wgrad_leaf(x, dy, s, want_bias=True, lp=ctx.lp, params=ctx.leaf_params)     # bias gradient from the same pass
                want_db = False
            elif ctx.hip_w:
                dw = conv2d_wgrad_leaf(x, dy, s, lp=ctx.lp, params=ctx.leaf_params)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        if want_db:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None, None




class _Conv1x1S2Fn(torch.autograd.Function):
    """Bias-free 1x1 / stride-2 downsample convolution of a BEV stage (custom_resnet_spc.conv1x1, odom_pred.py:404-406):
    forward and data gradient on rslo_conv2d_fwd_s2 / _dgrad_s2; the weight gradient is a plain [cout x cin] GEMM over
    the sampled pixels (library bmm on the subsampled input: 1.1 GFLOP, no layout transposes)."""

    @staticmethod
    def forward(ctx, x, w):
        from rslo_amd import capi
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ws = getattr(w, "_hip_split", None)
        if ws is not None and (ws[2] != w._version or ws[3] != w.data_ptr()):
            ws = None
        ctx.ws_t = ws[1] if ws is not None else capi.conv2d_wsplit_k(w, True)
        return capi.conv2d_fwd_s2(x, ws[0] if ws is not None else capi.conv2d_wsplit_k(w, False), w.shape[0], 1)

    @staticmethod
    def backward(ctx, dy):
        from rslo_amd import capi
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = capi.conv2d_dgrad_s2(dy, ctx.ws_t, w.shape[1], x.shape[2], x.shape[3], 1)
        if ctx.needs_input_grad[1]:
            dw = conv1x1s2_wgrad_leaf(x, dy, w.shape, params=(w,))
        return dx, dw


class _Conv1x1Fn(torch.autograd.Function):
    """1x1 / stride-1 output convolutions with at most 8 output channels through rslo_conv1x1_* (csrc/conv2d.hip)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        from rslo_amd import capi
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.leaf_params = (w,) if bias is None else (w, bias)
        return capi.conv1x1_fwd(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        from rslo_amd import capi
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = capi.conv1x1_dgrad(dy, w)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            # leaf work like the 3x3 layers' weight gradients (round 6: five launch pairs per step off the training stream)
            from rslo_amd import streams
            dw, db = streams.leaf(lambda: capi.conv1x1_wgrad(x, dy, want_bias=ctx.has_bias), (x, dy), ctx.leaf_params)
        return dx, dw, db


def presplit(root):
    """Refresh the split-bf16 operands of every eligible Conv2d under `root` (both orientations) in ONE launch; the
    layers pick them up while the weights' version counters are unchanged (i.e. until the next optimizer step)."""
    if "f" not in HIP_PASSES and "d" not in HIP_PASSES:
        return
    from rslo_amd import capi
    ent = root.__dict__.get("_hip_conv2d_plan")        # (plan, weights, views), kept on the module itself
    if ent is None or any(w.data_ptr() != p for w, p in zip(ent[1], ent[0]["ptrs"])):
        ws = [m.weight for m in root.modules()
              if isinstance(m, Conv2d) and m.hip_wgrad and m._kind() in ("3x3", "1x1s2") and m.weight.is_cuda
              and m.weight.dtype == torch.float32 and m.in_channels % 32 == 0 and m.out_channels % 32 == 0]
        if not ws:
            return
        plan, views = capi.conv2d_wsplit_many(ws)
        ent = (plan, ws, views)
        root.__dict__["_hip_conv2d_plan"] = ent
    capi.conv2d_wsplit_run(ent[0])
    for w, (f, t) in zip(ent[1], ent[2]):       # kept on the parameter object: (fwd, dgrad, version, storage) at split time
        w._hip_split = (f, t, w._version, w.data_ptr())


def presplit_early(root, device, side):
    """presplit(root) on the side stream `side`, ordered behind everything issued so far (the optimizer step that changed the
    weights, the previous backward's readers of the operands).  The caller's stream waits for the returned event right
    before the first layer that uses the operands: the split (103 us for the BEV head: 190 MB) then runs beside the sparse
    encoder's forward instead of in front of the head.  RSLO_PRESPLIT_EARLY=0: not used."""
    if (os.environ.get("RSLO_PRESPLIT_EARLY", "1") == "0" or torch.device(device).type != "cuda"
            or not getattr(root, "consumes_presplit_event", False)):      # only heads whose forward waits for the event
        return None
    # `side`: a stream the caller already owns.  NOT a new one: HIP maps streams onto a few hardware queues, and one more
    # stream put the training stream on a queue with a busy side stream in the distributed step (23 ms instead of 12)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        presplit(root)
        ev = torch.cuda.Event()
        ev.record(side)
    root.__dict__["_presplit_event"] = ev
    return ev


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose weight gradient runs on rslo_conv2d_wgrad when the layer is 3x3 / padding 1 / stride 1 or 2 /
    ungrouped on a CUDA fp32 tensor and the shape is supported; anything else is plain nn.Conv2d."""
    hip_wgrad = True

    def _kind(self):
        """Static part of the dispatch (module configuration only): "1x1", "3x3" or None.  Cached per configuration."""
        key = (self.kernel_size, self.stride, self.padding, self.dilation, self.groups, self.padding_mode,
               self.in_channels, self.out_channels)
        cached = self.__dict__.get("_hip_kind")
        if cached is not None and cached[0] == key:
            return cached[1]
        kind = None
        if self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros":
            if (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
                    and self.out_channels <= 8 and self.in_channels <= 256):
                kind = "1x1"
            elif self.kernel_size == (3, 3) and self.padding == (1, 1) and self.stride in ((1, 1), (2, 2)):
                kind = "3x3"
            elif (HIP_STRIDE2 and self.kernel_size == (1, 1) and self.stride == (2, 2) and self.padding == (0, 0) and self.bias is None
                  and self.in_channels % 32 == 0 and self.out_channels % 32 == 0):
                kind = "1x1s2"
        self.__dict__["_hip_kind"] = (key, kind)
        return kind

    def _eligible(self, x):
        ok = getattr(self, "_hip_ok", None)
        key = (x.shape[2], x.shape[3], HIP_PASSES)
        if ok is None or ok[0] != key:
            from rslo_amd import capi
            w_ok = "w" in HIP_PASSES and capi.conv2d_wgrad_supported(self.in_channels, self.out_channels, key[0],
                                                                     key[1], self.stride[0])
            fd_ok = ("f" in HIP_PASSES or "d" in HIP_PASSES) and (
                capi.conv2d_fwd_supported(self.in_channels, self.out_channels, key[0], key[1]) if self.stride == (1, 1)
                else (HIP_STRIDE2 and self.bias is None and capi.conv2d_s2_supported(self.in_channels, self.out_channels, 3)))
            ok = self._hip_ok = (key, w_ok, fd_ok)
        return ok[1] or ok[2]

    def _conv_forward(self, input, weight, bias):
        if (self.hip_wgrad and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
                and torch.is_grad_enabled() and weight.requires_grad):
            kind = self._kind()
            if kind == "1x1" and "w" in HIP_PASSES:
                return _Conv1x1Fn.apply(input, weight, bias)
            if kind == "1x1s2" and "f" in HIP_PASSES and "d" in HIP_PASSES and not _low_precision():
                return _Conv1x1S2Fn.apply(input, weight)
            if kind == "3x3" and self._eligible(input):
                return _Conv3x3Fn.apply(input, weight, bias, self.stride[0], self._hip_ok[2], self._hip_ok[1])
        return super()._conv_forward(input, weight, bias)
