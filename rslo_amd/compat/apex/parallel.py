import os

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
        if self.prev is not None:       # nested: the outermost context applies everything at once
            self.prev.extend(pending)
            return False
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


class isolated_batch_counts:
    """Inside this context a defer_batch_counts() block is an OUTERMOST one whatever encloses the caller: its increments are
    applied at ITS exit.  rslo_amd/headgraph.py captures the head's forward into a hipGraph while the network's own
    defer_batch_counts() is open; the increments have to be launches of the capture, not of the enclosing context."""

    def __enter__(self):
        global _pending_counts
        self.prev, _pending_counts = _pending_counts, None
        return self

    def __exit__(self, *exc):
        global _pending_counts
        _pending_counts = self.prev
        return False


def count_batch(bn):
    """`bn.num_batches_tracked += 1`, deferred when inside defer_batch_counts()."""
    if bn.num_batches_tracked is None:
        return
    if _pending_counts is not None:
        _pending_counts.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


# "1" (default): the fused kernels of csrc/bn2d.hip run on any number of ranks -- one rank: statistics slices + apply,
# two launches per direction (26 us per layer fwd + bwd against 35 us for MIOpen's BatchNorm + separate ReLU / add
# kernels); more ranks: slices -> finish -> all-reduce -> apply, replacing torch SyncBatchNorm's statistics gather +
# separate activation / add.  "auto": fused only with more than one rank, "0": never (library kernels).
FUSED_BN = os.environ.get("RSLO_FUSED_BN", "1")


def _world(group):
    if not dist.is_initialized():
        return 1
    return dist.get_world_size(group) if group is not None else dist.get_world_size()


_HP_GROUP = {"tried": False, "group": None}


def exchange_group(group):
    """The process group the SyncBN statistics exchanges run on.  A step issues ~90 of them (one per normalisation layer
    and direction), each a latency-bound all-reduce of a few hundred bytes between two dependent kernels while three other
    HIP streams (structure plan, covariance branch, weight gradients) keep the GPU's queues busy.  With the default group
    (`process_group=None`, what convert_syncbn_model passes) and the NCCL / RCCL backend they go to a dedicated group whose
    communication stream has HIGH priority, so the exchange kernel is not queued behind the side streams' work; the group
    is created on first use (a collective call: every rank reaches its first SyncBN forward at the same point).
    OPT-IN (RSLO_SYNCBN_HP_GROUP=1): a second communicator is one more HIP stream, and the step already runs five on
    the runtime's 4-8 hardware queues -- one stream too many measurably serialises the training stream behind a side
    stream (23 ms instead of 12 ms per step on one GPU), and this cannot be checked at N > 1 on a one-GPU box.
    Default, another backend, or an explicit process_group: unchanged."""
    if group is not None or os.environ.get("RSLO_SYNCBN_HP_GROUP", "0") != "1":
        return group
    if _HP_GROUP.get("world") is not dist.group.WORLD:          # a re-initialised default group: the cached one is dead
        _HP_GROUP.update(tried=False, group=None, world=dist.group.WORLD)
    if not _HP_GROUP["tried"]:
        _HP_GROUP["tried"] = True
        if dist.get_backend() == "nccl":
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            _HP_GROUP["group"] = dist.new_group(backend="nccl", pg_options=opts)
    return _HP_GROUP["group"]


# RSLO_FORCE_SYNCBN_PATH=1: a ONE-rank process group runs the multi-rank code path (statistics that leave the kernel /
# meet the peers inside it, a world-size-1 peer comm) -- what every rank of an N > 1 job executes per layer, measurable on
# a one-GPU box (bench.py reports it as config.multirank_path_ms_per_step).  Never on by default.
FORCE_MULTI = os.environ.get("RSLO_FORCE_SYNCBN_PATH", "0") == "1"
# "1" (default): the maps a workgroup holds in registers exchange inside ONE kernel per direction (rslo_bn2d_fwd_peer), on
# either peer transport (two processes on one GPU, 256 channels at 12x22: 16 us against 33 us on the device transport,
# 22 against 33 on the host segment); "0": statistics kernel -> exchange kernel -> apply kernel everywhere (round 4)
FUSED_PEER_BN = os.environ.get("RSLO_SYNCBN_FUSED_PEER", "1") != "0"


# The single-launch / three-launch choice of a layer must come out THE SAME on every rank (a rank on the other path would wait
# in another slot region until the timeout).  It depends on the per-rank batch through N * HW <= 17408: with equal per-rank
# batches -- what the reference's DistributedSampler + drop-less DataLoader deliver, train_hdf5.py:300-330 -- every rank decides
# alike from its own shape.  Where per-rank batch sizes CAN differ (a short last batch on some ranks), set
# RSLO_SYNCBN_MAX_BATCH to the largest per-rank batch: the choice is then made for that size on every rank, whatever it holds.
MAX_BATCH = int(os.environ.get("RSLO_SYNCBN_MAX_BATCH", "0"))


def _peer_shape_ok(x):
    from rslo_amd import capi
    return capi.bn2d_peer_supported(max(int(x.shape[0]), MAX_BATCH), x.shape[1], x.shape[2] * x.shape[3])


def _fused_peer_comm(group):
    """The peer comm when this layer's exchange happens inside the BatchNorm kernel, else None."""
    if not FUSED_PEER_BN:
        return None
    from rslo_amd import peer
    comm = peer.comm_for(group)
    if comm is not None and comm.shared_device and not FORCE_FUSED_ON_SHARED_DEVICE:
        return None          # ranks sharing one GPU: see PeerComm.shared_device
    return comm


FORCE_FUSED_ON_SHARED_DEVICE = False      # tests/test_peer_exchange.py sets it for its small layers


def _multi(world):
    return world > 1 or (FORCE_MULTI and dist.is_available() and dist.is_initialized())


def _sum_over_ranks(t, group):
    """The SyncBN statistics exchange: a same-stream kernel when all ranks share this host and the tensor fits one
    exchange (rslo_amd.peer.all_reduce_: float64, <= 1024 elements), else the all-reduce on exchange_group(group)."""
    from rslo_amd import peer
    peer.all_reduce_(t, group, fallback_group=exchange_group(group))


def fused_bn_forward(bn, x, res, weight, bias, slope, group, world):
    """y = act(BN_train(x) + res) through rslo_bn2d_* (x, res contiguous).  -> y, mean, invstd, cnt_all (the element
    count over all ranks as a device scalar; None on one rank).  Updates bn's running statistics."""
    from rslo_amd import capi
    track = bn.track_running_stats and bn.running_mean is not None
    mom = bn.momentum          # fusable() leaves momentum=None (cumulative average) to the unfused path
    if _multi(world):
        comm = _fused_peer_comm(group)
        if comm is not None and _peer_shape_ok(x):
            # one launch: the workgroup of a channel meets its peers on the other ranks between its sums and its apply
            return capi.bn2d_fwd_peer(comm, x, res, weight, bias, bn.running_mean if track else None,
                                      bn.running_var if track else None, mom, bn.eps, slope)
        stats = capi.bn2d_stats(x)
        _sum_over_ranks(stats, group)
        y, mean, invstd = capi.bn2d_apply(x, res, stats, weight, bias, bn.running_mean if track else None,
                                          bn.running_var if track else None, mom, bn.eps, slope)
        return y, mean, invstd, stats[-1:]        # ranks may hold different batch sizes: the all-reduced count
    # one rank: no exchange, the apply kernel adds the slice partials itself (two launches)
    y, mean, invstd = capi.bn2d_fwd_local(x, res, weight, bias, bn.running_mean if track else None,
                                          bn.running_var if track else None, mom, bn.eps, slope)
    return y, mean, invstd, None


def fused_bn_backward(gy, y, x, weight, mean, invstd, cnt_all, slope, has_res, affine, group, world):
    """-> dx, dres, dgamma, dbeta (dgamma / dbeta are this rank's sums: data parallel averages them afterwards)."""
    from rslo_amd import capi
    has_act = slope != 1.0
    if cnt_all is None:          # the forward ran the one-rank path
        return capi.bn2d_bwd_local(gy, y, x, weight, mean, invstd, slope, has_act, has_res, want_affine=affine)
    comm = _fused_peer_comm(group)      # the same condition the forward took: both directions of a layer use one path
    if comm is not None:
        if _peer_shape_ok(x):
            return capi.bn2d_bwd_peer(comm, gy, y, x, weight, mean, invstd, cnt_all, slope, has_act, has_res, want_affine=affine)
    red, dgamma, dbeta = capi.bn2d_bwd_reduce(gy, y, x, mean, invstd, slope, has_act, want_affine=affine)
    _sum_over_ranks(red, group)
    # the count all-reduced in the forward pass, read on the device: exact for uneven per-rank batches, no host sync
    dx, dres = capi.bn2d_bwd_apply(gy, y, x, weight, mean, invstd, red, 0.0, slope, has_act, has_res, count_dev=cnt_all)
    return dx, dres, dgamma, dbeta


class _FusedBNActFn(torch.autograd.Function):
    """y = act(BN_train(x) + residual) through rslo_bn2d_* (rslo_amd/csrc/bn2d.hip).  Statistics are over all ranks of
    `group`: the per-channel sums are all-reduced between the two kernels of each direction (skipped on one rank)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn, slope, group):
        x = x.contiguous()
        res = None if residual is None else residual.contiguous()
        world = _world(group)
        y, mean, invstd, cnt_all = fused_bn_forward(bn, x, res, weight, bias, slope, group, world)
        ctx.save_for_backward(x, y if slope != 1.0 else None, weight, mean, invstd, cnt_all)
        ctx.meta = (slope, group, world, residual is not None, weight is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, mean, invstd, cnt_all = ctx.saved_tensors
        slope, group, world, has_res, affine = ctx.meta
        dx, dres, dgamma, dbeta = fused_bn_backward(gy.contiguous(), y, x, weight, mean, invstd, cnt_all, slope, has_res,
                                                    affine, group, world)
        return dx, dgamma, dbeta, dres, None, None, None


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

    def fusable(self, x):
        """The fused kernels take training-mode float32 NCHW tensors on the GPU (any number of ranks)."""
        if FUSED_BN == "0" or self.momentum is None or not (self.training and isinstance(x, torch.Tensor) and x.is_cuda
                                                            and x.dim() == 4 and x.dtype == torch.float32):
            return False
        return FUSED_BN != "auto" or _world(self.process_group) > 1

    def forward(self, x, act_slope=None, residual=None):
        """act_slope / residual: optional fused epilogue  y = act(bn(x) + residual)  (act_slope 0 = ReLU, None = no
        activation); without the fused kernels the same thing is computed with separate ops."""
        if self.fusable(x):
            if self.track_running_stats:
                count_batch(self)
            return _FusedBNActFn.apply(x, self.weight, self.bias, residual, self,
                                       1.0 if act_slope is None else float(act_slope), self.process_group)
        y = self._plain_forward(x)
        if residual is not None:
            y = y + residual
        if act_slope is not None:
            y = F.leaky_relu(y, act_slope) if act_slope != 0.0 else F.relu(y)
        return y

    def _plain_forward(self, x):
        single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if single or not self.training:
            if self.training and self.track_running_stats:
                count_batch(self)
            use_batch = self.training or not self.track_running_stats
            mom = self.momentum
            if mom is None:         # torch: cumulative moving average, factor 1 / num_batches_tracked (already counted)
                n = int(self.num_batches_tracked) if self.num_batches_tracked is not None else 0
                mom = 1.0 / n if (self.training and n > 0) else 0.0
            return F.batch_norm(x, self.running_mean if self.track_running_stats else None,
                                self.running_var if self.track_running_stats else None, self.weight, self.bias,
                                use_batch, mom, self.eps)
        return super().forward(x)


def convert_syncbn_model(module, process_group=None, channel_last=False):
    return nn.SyncBatchNorm.convert_sync_batchnorm(module, process_group)


class DistributedDataParallel(nn.Module):
    """apex.parallel.DistributedDataParallel as the driver uses it (train_hdf5.py:463 `net_parallel =
    apex.parallel.DistributedDataParallel(net)`; forward through the wrapper, `.module` underneath), on the exchange this
    path is built around: parameters and buffers broadcast from rank 0 at construction, gradients AVERAGED over the ranks
    (apex: gradient_average=True) by rslo.utils.distributed_utils -- for the odometry network the BEV head's ~11 M
    gradients leave as one flat asynchronous all-reduce when the head's backward is done (OverlappedGradientExchange),
    the rest as a second flat bucket at the end of the pass; any other module: one flat bucket (average_gradients).
    Like apex, the exchange finishes inside backward(): an autograd-engine callback queued when the first output
    gradient arrives, so `scaled_loss.backward()` returns with reduced gradients and the driver's next statement
    (`clip_grad_norm_`, train_hdf5.py:671) sees them.  torch DDP with find_unused_parameters=True (77 of the 290
    parameter tensors never receive a gradient) would cost ~8 ms of host time per step here; message_size /
    delay_allreduce are accepted and have nothing to tune."""

    def __init__(self, module, message_size=10000000, delay_allreduce=False, shared_param=None,
                 allreduce_trigger_params=None, retain_allreduce_buffers=False, allreduce_always_fp32=False,
                 num_allreduce_streams=1, allreduce_communicators=None, gradient_average=True,
                 gradient_predivide_factor=1.0, gradient_average_split_factor=None, prof=False):
        super().__init__()
        if gradient_predivide_factor != 1.0 or not gradient_average:
            raise NotImplementedError("apex DDP stand-in: gradients are averaged over the ranks (the reference's setting)")
        self.module = module
        self._exchange = None
        self._queued = False
        if dist.is_available() and dist.is_initialized():
            from rslo.utils import distributed_utils as du
            du.broadcast_params(module, 0)
            early = getattr(module, "odom_predictor", None)
            if early is not None and os.environ.get("RSLO_OVERLAP_GRADS", "1") != "0":
                self._exchange = du.OverlappedGradientExchange(module, early, mean=True, module_hook=False)
                module.__dict__["_grad_exchange"] = self._exchange       # the network marks the head / encoder boundary

    def _finish(self):
        from rslo.utils import distributed_utils as du
        self._queued = False
        import sys
        st = sys.modules.get("rslo_amd.streams")
        if st is not None:        # weight gradients still on the leaf stream (its own end-of-pass callback was queued after
            st.join()             # this one): the buckets are built on the issuing stream, behind them
            st.flush_deferred()   # ... and the collected sparse weight-gradient reduces of the issuing stream
        if self._exchange is not None:
            self._exchange.finish()
        else:
            du.average_gradients(self.module, mean=True)
        pr = sys.modules.get("rslo_amd.peer")
        if pr is not None:        # a SyncBN exchange that timed out poisoned its statistics with NaN: an error within the
            pr.check_all()        # step that follows it (pinned-memory read, no synchronisation), never a silent one

    def _on_first_grad(self, grad):
        if not self._queued:
            self._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        return None

    def forward(self, *inputs, **kwargs):
        out = self.module(*inputs, **kwargs)
        self._queued = False      # (a backward pass that raised before its callback ran must not silence the next one)
        if torch.is_grad_enabled() and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            ts = out.values() if isinstance(out, dict) else (out if isinstance(out, (list, tuple)) else (out,))
            for t in ts:
                if torch.is_tensor(t) and t.requires_grad:
                    t.register_hook(self._on_first_grad)
        return out
