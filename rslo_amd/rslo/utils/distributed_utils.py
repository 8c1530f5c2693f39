"""Data-parallel helpers of the hot path (reference: rslo/utils/distributed_utils.py:53-71,238-314).

The path shards by frame pair: every rank runs the whole network on its own samples and the only
mandatory exchange is the gradient all-reduce.  On MI355X that is RCCL over xGMI through torch.distributed
(backend "nccl"); ring all-reduce is per-link bound, so the 48 MB of gradients go as ONE flat bucket instead of
the reference's per-parameter all-reduce loop (213 latency-bound messages)."""
import math

import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


def average_gradients(model, bucket=True):
    """All-reduce (mean) the gradients of `model` across ranks: one flat fp32 bucket."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    params = [p for p in model.parameters() if p.requires_grad and p.grad is not None]
    if not params:
        return
    world = dist.get_world_size()
    if not bucket:
        for p in params:
            dist.all_reduce(p.grad.data)
            p.grad.data.div_(world)
        return
    grads = [p.grad for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat.div_(world)
    torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])


def broadcast_params(model, src=0):
    """Rank `src` -> all, one flattened broadcast per dtype (reference: per-tensor loop, :68-71)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    by_dtype = {}
    for t in model.state_dict().values():
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


class DistributedGivenIterationSamplerEpoch(Sampler):
    """Index shard of one rank for a fixed number of iterations: every rank shuffles the dataset with the SAME
    seed, repeats it to cover total_iter * batch_size * world_size samples and takes its contiguous slice
    [total_size * rank, total_size * (rank + 1)) (reference :238-314; seed 7, :270).  Yields (index, seed)."""

    def __init__(self, dataset_len, total_iter, batch_size, world_size=None, rank=None, last_iter=-1, seed=7):
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        assert rank < world_size
        self.dataset_len = dataset_len
        self.total_iter, self.batch_size = total_iter, batch_size
        self.world_size, self.rank, self.last_iter, self.seed = world_size, rank, last_iter, seed
        self.total_size = total_iter * batch_size
        self.indices = self._gen()

    def _gen(self):
        g = torch.Generator().manual_seed(self.seed)
        all_size = self.total_size * self.world_size
        reps = int(math.ceil(all_size / self.dataset_len))
        idx = torch.cat([torch.randperm(self.dataset_len, generator=g) for _ in range(reps)])[:all_size]
        beg = self.total_size * self.rank
        return idx[beg:beg + self.total_size].tolist()

    def __iter__(self):
        start = (self.last_iter + 1) * self.batch_size
        return iter([(i, self.seed) for i in self.indices[start:]])

    def __len__(self):
        return self.total_size - (self.last_iter + 1) * self.batch_size
