import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

ReduceOp = dist.ReduceOp


_pending_counts = None


class defer_batch_counts:
    """Inside this context the single-rank SyncBatchNorm layers do not launch their own `num_batches_tracked += 1`
    (45 one-element kernels per forward of the BEV head); the increments are collected and applied at exit with one
    multi-tensor add per distinct increment.  Same buffer values after the forward."""

    def __enter__(self):
        global _pending_counts
        self.prev, _pending_counts = _pending_counts, []
        return self

    def __exit__(self, *exc):
        global _pending_counts
        pending, _pending_counts = _pending_counts, self.prev
        by_id = {}
        for t in pending:
            ent = by_id.setdefault(id(t), [t, 0])
            ent[1] += 1
        for inc in sorted({c for _, c in by_id.values()}):
            ts = [t for t, c in by_id.values() if c == inc]
            if ts[0].is_cuda:
                torch._foreach_add_(ts, inc)
            else:
                for t in ts:
                    t.add_(inc)
        return False


def count_batch(bn):
    """`bn.num_batches_tracked += 1`, deferred when inside defer_batch_counts()."""
    if bn.num_batches_tracked is None:
        return
    if _pending_counts is not None:
        _pending_counts.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


class SyncBatchNorm(nn.SyncBatchNorm):
    """apex.parallel.SyncBatchNorm signature on torch's RCCL-backed SyncBatchNorm.  With a single
    process (or no process group) it is exactly BatchNorm (biased variance for normalisation,
    unbiased for the running estimate), and then also works on any device."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None, channel_last=False, fuse_relu=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats, process_group=process_group)
        self.channel_last = channel_last
        self.fuse_relu = fuse_relu

    def forward(self, x):
        single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if single or not self.training:
            if self.training and self.track_running_stats:
                count_batch(self)
            use_batch = self.training or not self.track_running_stats
            return F.batch_norm(x, self.running_mean if self.track_running_stats else None,
                                self.running_var if self.track_running_stats else None, self.weight, self.bias,
                                use_batch, self.momentum if self.momentum is not None else 0.0, self.eps)
        return super().forward(x)


def convert_syncbn_model(module, process_group=None, channel_last=False):
    return nn.SyncBatchNorm.convert_sync_batchnorm(module, process_group)


class DistributedDataParallel(nn.parallel.DistributedDataParallel):
    """apex DDP call signature (train_hdf5.py:463) on torch DDP over RCCL.  77 of the 290 parameter
    tensors never receive a gradient, hence find_unused_parameters."""

    def __init__(self, module, message_size=10000000, delay_allreduce=False, **kwargs):
        dev = next(module.parameters()).device
        ids = [dev.index] if dev.type == "cuda" else None
        super().__init__(module, device_ids=ids, find_unused_parameters=True,
                         bucket_cap_mb=max(1, int(message_size * 4 / 2 ** 20)))
