"""Normalisation variants of the registry that the shipped configuration does not select (SURVEY.md 8f-4;
reference: rslo/layers/normalization.py:11-251).

  MaskSyncBatchNorm         batch statistics over the OCCUPIED cells only: the input is zeroed outside the mask and the
                            plain means are rescaled by (cells / occupied cells) -- note the reference rescales the
                            second moment by the SQUARE of that factor (normalization.py:76-77), kept as is.
  SemiGlobalSyncBatchNorm   normalises with the RUNNING statistics, which follow the batch statistics with a per-channel
                            momentum adapted from the relative drift of a probe average (normalization.py:124-139).

Both hand (mean, var) to apex's `SyncBatchnormFunction`.  apex is not in this image and not in the reference tree, so
its published Python fallback is restated here (`_SyncBNApply`, parity UNPINNED for that function):
  y = (x - mean) / sqrt(var + eps) * w + b;
  dx = (dy - mean(dy) - (x - mean) / (var + eps) * mean(dy (x - mean))) / sqrt(var + eps) * w   (means all-reduced),
  dw = sum(dy (x - mean) / sqrt(var + eps)),  db = sum(dy).
Plain torch ops: these variants are off the hot path.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from apex.parallel import SyncBatchNorm


def _world(group):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


class _SyncBNApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mean, var, eps, group, world):
        shape = [1, -1] + [1] * (x.dim() - 2)
        ctx.save_for_backward(x, weight, mean, var)
        ctx.meta = (eps, group, world, shape)
        y = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps)
        if weight is not None:
            y = y * weight.view(shape)
        if bias is not None:
            y = y + bias.view(shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, var = ctx.saved_tensors
        eps, group, world, shape = ctx.meta
        dims = [d for d in range(x.dim()) if d != 1]
        xmu = x - mean.view(shape)
        inv = 1.0 / torch.sqrt(var + eps)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            mean_dy = gy.mean(dims)
            mean_dy_xmu = (gy * xmu).mean(dims)
            if world > 1:
                dist.all_reduce(mean_dy, group=group)
                dist.all_reduce(mean_dy_xmu, group=group)
                mean_dy, mean_dy_xmu = mean_dy / world, mean_dy_xmu / world
            gx = (gy - mean_dy.view(shape) - xmu / (var.view(shape) + eps) * mean_dy_xmu.view(shape)) * inv.view(shape)
            if weight is not None:
                gx = gx * weight.view(shape)
        if weight is not None and ctx.needs_input_grad[1]:
            gw = (gy * xmu * inv.view(shape)).sum(dims)
        if ctx.needs_input_grad[2]:
            gb = gy.sum(dims)
        return gx, gw, gb, None, None, None, None, None


def _batch_moments(x, group, scale=1.0):
    """Per-channel E[x], E[x^2] over batch and space (x `scale`, x `scale`^2), averaged over the ranks; m = elements."""
    flat = x.transpose(0, 1).reshape(x.shape[1], -1)
    m = float(flat.shape[1])
    mean = flat.mean(1) * scale
    sqr = flat.pow(2).mean(1) * scale ** 2
    world = _world(group)
    if world > 1:
        dist.all_reduce(mean, group=group)
        dist.all_reduce(sqr, group=group)
        mean, sqr, m = mean / world, sqr / world, m * world
    return mean, sqr - mean.pow(2), m, world


class MaskSyncBatchNorm(SyncBatchNorm):
    def forward(self, input):
        x, mask = input
        mask = (mask[:, 0:1] > 0).to(x.dtype).detach()
        x = x * mask
        if not self.training and self.track_running_stats:
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
        valid = mask.sum()
        total = x.shape[0] * int(torch.tensor(x.shape[2:]).prod())
        rectifier = total / (valid + 1e-3)
        self.num_batches_tracked += 1
        with torch.no_grad():
            mean, var, m, world = _batch_moments(x, self.process_group, rectifier)
            if self.running_mean is not None:
                self.running_mean = self.momentum * mean + (1 - self.momentum) * self.running_mean
            if self.running_var is not None:
                self.running_var = m / (m - 1) * self.momentum * var + (1 - self.momentum) * self.running_var
        return _SyncBNApply.apply(x, self.weight, self.bias, mean, var, self.eps, self.process_group, world)


class SemiGlobalSyncBatchNorm(SyncBatchNorm):
    def __init__(self, num_features, eps=1e-05, momentum=0.1, affine=True, track_running_stats=True, process_group=None,
                 channel_last=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats, process_group=process_group)
        base = momentum
        self.momentum, self.avg_runing, self.dyn_mom, self.iter_cnt = 0.9, False, True, 0
        self.register_buffer("mean_dyn_mom", torch.full((num_features,), float(base)))
        self.register_buffer("var_dyn_mom", torch.full((num_features,), float(base)))
        self.register_buffer("running_mean_g2", torch.ones(num_features))
        self.register_buffer("running_var_g2", torch.ones(num_features))
        self.register_buffer("running_mean_probe", torch.zeros(num_features))
        self.register_buffer("running_var_probe", torch.ones(num_features))

    def update_momentum_and_g2(self, dyn_mom, running_g2, running_probe, val, beta=0.1):
        old = running_probe
        running_probe = (1 - beta) * running_probe + beta * val
        diff = ((running_probe - old) / old) ** 2
        running_g2 = torch.clamp((1 - beta) * running_g2 + beta * diff, max=self.momentum ** 2, min=0)
        dyn_mom = 1 - (1 - self.momentum) / (1 - self.momentum + torch.sqrt(running_g2) + 1e-9)
        return dyn_mom, running_g2, running_probe

    def forward(self, x):
        if not self.training and self.track_running_stats:
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
        self.num_batches_tracked += 1
        with torch.no_grad():
            mean, var, m, world = _batch_moments(x, self.process_group)
            if self.iter_cnt < 1 / self.momentum:
                self.iter_cnt += 1
            if self.running_mean is not None:
                mom = self.mean_dyn_mom
                self.running_mean = mom * mean + (1 - mom) * self.running_mean.detach()
                self.mean_dyn_mom, self.running_mean_g2, self.running_mean_probe = self.update_momentum_and_g2(
                    self.mean_dyn_mom, self.running_mean_g2, self.running_mean_probe, mean)
            if self.running_var is not None:
                mom = self.var_dyn_mom
                self.running_var = m / (m - 1) * mom * var + (1 - mom) * self.running_var.detach()
                self.var_dyn_mom, self.running_var_g2, self.running_var_probe = self.update_momentum_and_g2(
                    self.var_dyn_mom, self.running_var_g2, self.running_var_probe, var)
        return _SyncBNApply.apply(x, self.weight, self.bias, self.running_mean.detach(), self.running_var.detach(),
                                  self.eps, self.process_group, world)


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
