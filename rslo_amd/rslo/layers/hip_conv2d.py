"""Conv2d of the BEV head with the weight gradient on the hand-written gfx950 kernel.

The reference builds its BEV encoder-decoder from torch.nn.Conv2d (rslo/models/odom_pred.py:65-134,398-426,
rslo/layers/MaskConv.py:20-73) and leaves all three convolution passes to cuDNN.  Here forward and data gradient stay
with the library (MIOpen's fp32 Winograd kernels), the weight gradient of every 3x3 / padding-1 layer whose shape the
kernel takes goes through rslo_conv2d_wgrad (csrc/conv2d.hip): one kernel + one reduce instead of MIOpen's
split-K igemm + 4 layout transposes + a zero fill, fixed summation order.  Same parameters, same state-dict keys."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        ctx.has_bias = bias is not None
        return F.conv2d(x, w, bias, stride, 1)

    @staticmethod
    def backward(ctx, dy):
        from rslo_amd import capi
        x, w = ctx.saved_tensors
        s = ctx.stride
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = capi.conv2d_wgrad(x, dy, s)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose weight gradient runs on rslo_conv2d_wgrad when the layer is 3x3 / padding 1 / stride 1 or 2 /
    ungrouped on a CUDA fp32 tensor and the shape is supported; anything else is plain nn.Conv2d."""
    hip_wgrad = True

    def _eligible(self, x):
        if not (self.hip_wgrad and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and torch.is_grad_enabled()
                and self.weight.requires_grad):
            return False
        if self.kernel_size != (3, 3) or self.padding != (1, 1) or self.dilation != (1, 1) or self.groups != 1 \
                or self.stride not in ((1, 1), (2, 2)) or self.padding_mode != "zeros":
            return False
        ok = getattr(self, "_hip_ok", None)
        key = (x.shape[2], x.shape[3])
        if ok is None or ok[0] != key:
            from rslo_amd import capi
            ok = self._hip_ok = (key, capi.conv2d_wgrad_supported(self.in_channels, self.out_channels, key[0], key[1],
                                                                 self.stride[0]))
        return ok[1]

    def _conv_forward(self, input, weight, bias):
        if self._eligible(input):
            return _Conv3x3Fn.apply(input, weight, bias, self.stride[0])
        return super()._conv_forward(input, weight, bias)
