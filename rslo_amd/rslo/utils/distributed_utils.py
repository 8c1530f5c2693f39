"""Data-parallel helpers of the training driver (reference: rslo/utils/distributed_utils.py:12-314).

The path shards by frame pair: every rank runs the whole network on its own samples; the only mandatory exchange
is the gradient all-reduce.  On MI355X that is RCCL over xGMI through torch.distributed (backend "nccl").  xGMI rings
are per-link bound and launch-latency dominated for small messages, so the ~12 M fp32 gradients (48 MB) travel as ONE
flat bucket instead of the reference's per-parameter loop (213 messages), and parameters are broadcast as one
flat buffer per dtype.

Semantics kept from the reference:
  * `average_gradients` SUMS (the driver divides the loss by the world size first, train_hdf5.py:666-669);
    `mean=True` folds the division in (used by bench.py);
  * the samplers reproduce the reference's index lists exactly (numpy global RNG seeded with 7,
    distributed_utils.py:212,270), so a resumed or multi-rank run sees the same samples.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.nn import Module
from torch.utils.data.sampler import Sampler


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def gradients_multiply(model, multiplier=1):
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    if grads:
        torch._foreach_mul_(grads, multiplier)


_HOST_GROUP = {}


def _host_group():
    """Group for HOST-side agreements (one flag per call, a presence vector when the flag is up).  Decisions that shape
    a collective -- which tensors a bucket holds -- must be identical on every rank and are taken on the host, so they
    travel over a CPU backend: the default group itself when it is gloo, else a gloo group created next to it on first
    use (all ranks reach this point together: they call average_gradients / finish() the same number of times).  A GPU
    all-reduce of the flag would have to be READ on the host, i.e. drain the training stream every step.  Returns
    None when no CPU group can be had; the agreement then falls back to the default group plus a host read."""
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    if _HOST_GROUP.get("world") is not dist.group.WORLD:       # first use, or the process group was destroyed and re-made
        _HOST_GROUP.clear()
        _HOST_GROUP["world"] = dist.group.WORLD
        try:
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
                # single-node launch: gloo would look its interface up through the host NAME, which need not resolve
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            _HOST_GROUP["g"] = dist.new_group(backend="gloo")
        except Exception:
            _HOST_GROUP["g"] = None
    return _HOST_GROUP["g"]


def _host_max(values, device):
    """Element-wise MAX of a small int list over all ranks, result on the host."""
    g = _host_group()
    if g is not None:
        t = torch.tensor(values, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=g)
        return t.tolist()
    t = torch.tensor(values, dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


def _agreed_grad_set(model, params):
    """Indices (into `params`) of the parameters that take part in the gradient exchange: every parameter that has
    received a gradient on ANY rank so far.  The set only grows.  Each call exchanges ONE flag over the host group
    ("I hold a gradient outside the set" / "I have no set yet"); when any rank raises it, all ranks re-agree in the
    same call through a presence vector -- so a parameter that first gets a gradient late (a loss branch that switches
    on with global_step, a mask that was empty on every rank when the set was first agreed) joins the bucket on all
    ranks at once instead of hanging the collective or being dropped.  The reference's per-parameter loop
    (distributed_utils.py:62-75) stays aligned because its frozen torch zero-fills gradients; here
    `zero_grad(set_to_none=True)` recomputes the non-None set from each step's graph, and a rank whose graph skips a
    parameter of the set contributes zeros."""
    return _agreed_grad_set_and_flag(model, params, 0)[0]


def _agreed_grad_set_and_flag(model, params, flag):
    """_agreed_grad_set + one more per-rank integer riding the same host exchange: -> (indices, max of `flag` over ranks)."""
    st = model.__dict__.setdefault("_rslo_grad_set", {"idx": None, "n": None})
    fresh = st["idx"] is None or st["n"] != len(params)
    chosen = set() if fresh else set(st["idx"])
    stray = any(p.grad is not None and i not in chosen for i, p in enumerate(params))
    dev = params[0].device
    redo, flag = _host_max([1 if (fresh or stray) else 0, int(flag)], dev)
    if redo:
        have = _host_max([1 if (p.grad is not None or i in chosen) else 0 for i, p in enumerate(params)], dev)
        st["idx"] = [i for i, h in enumerate(have) if h > 0]
        st["n"] = len(params)
    return st["idx"], flag


def _reduce_bucket(sel, mean, async_op=False):
    """Flat all-reduce of the gradients of `sel` (zeros where this rank has none)."""
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in sel]
    flat = torch.cat([g.reshape(-1) for g in grads])
    work = dist.all_reduce(flat, async_op=async_op)
    return grads, flat, work


def _scatter_bucket(sel, grads, flat, mean):
    if mean:
        flat.div_(dist.get_world_size())
    torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])
    for p, g in zip(sel, grads):
        if p.grad is None:
            p.grad = g


def average_gradients(model, bucket=True, mean=False):
    """All-reduce the gradients of `model` across ranks as one flat fp32 bucket (sum; mean=True divides by the
    world size).  The bucket covers the agreed set of gradient-carrying parameters (same size on every rank by
    construction, see _agreed_grad_set); a rank whose graph skipped one of them this step contributes zeros and
    receives the sum."""
    if not _active():
        return
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return
    idx = _agreed_grad_set(model, params)
    if not idx:
        return
    sel = [params[i] for i in idx]
    if not bucket:
        world = dist.get_world_size()
        for p in sel:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            dist.all_reduce(g)
            if mean:
                g.div_(world)
            p.grad = g
        return
    grads, flat, _ = _reduce_bucket(sel, mean)
    _scatter_bucket(sel, grads, flat, mean)


class OverlappedGradientExchange:
    """average_gradients in two pieces, the first one overlapped with the rest of backward.

    The network's backward runs loss -> BEV head -> encoder; the head holds ~11 M of the 12 M parameters and its
    gradients are complete when the gradient w.r.t. the head's input (the BEV map) has been produced, while ~3 ms of
    encoder backward are still to run.  `install(model, early)` puts a backward hook on the module `early`
    (`net.odom_predictor`): when it fires, the gradients of that module's parameters (the agreed subset) go out as ONE
    flat asynchronous all-reduce; `finish()` -- called where the driver calls average_gradients -- waits for it and
    reduces the remaining tensors as a second flat bucket.  Same sums as average_gradients (bit-identical on two ranks:
    each element is still the sum of the same two numbers).  xGMI rings are per-link bound: two large messages, not 213.

    The SEQUENCE of collectives is a function of the agreed set alone, never of one rank's graph: whenever a set
    exists, every rank sends [early subset, rest] -- from the hook if it fires, else from finish() -- with zeros for
    gradients that rank does not hold (a branch of the head skipped on one rank, a loss that did not reach the head)."""

    def __init__(self, model, early, mean=False, module_hook=True):
        """module_hook=False: the caller marks the boundary itself with watch(tensor) every forward (modules whose inputs
        or outputs are lists / dicts, like the BEV head: the hook sits on the BEV tensor the head consumes)."""
        self.model, self.mean = model, mean
        self.early_ids = {id(p) for p in early.parameters() if p.requires_grad}
        self.pending = None
        self.handle = early.register_full_backward_hook(self._on_early_done) if module_hook else None

    def remove(self):
        if self.handle is not None:
            self.handle.remove()

    def watch(self, tensor):
        """The gradient w.r.t. `tensor` is complete exactly when the early module's backward is: launch the bucket then."""
        if tensor.requires_grad and _active():
            tensor.register_hook(lambda g: self._on_early_done(None, None, None))
        return tensor

    def _early_subset(self, params):
        """The early bucket's tensors under the set agreed by EARLIER finish() calls (identical on every rank); empty
        before the first agreement -- at hook time the later modules have no gradients yet, so an agreement taken
        there would be wrong, and the first step goes out in one piece at finish()."""
        st = self.model.__dict__.get("_rslo_grad_set")
        if not st or st.get("idx") is None or st.get("n") != len(params):
            return []
        return [params[i] for i in st["idx"] if id(params[i]) in self.early_ids]

    def _launch_early(self, params):
        sel = self._early_subset(params)
        if sel:
            import sys
            conv = sys.modules.get("rslo.layers.hip_conv2d")
            if conv is not None:            # weight gradients issued on the leaf stream: this stream waits for them first
                conv.join_leaf_stream()
            # a parameter whose uses all lie behind the watched tensor has its gradient by now (the engine runs a ready
            # AccumulateGrad ahead of everything else); a missing one goes out as zeros.  A parameter that is ALSO used in
            # front of the watched tensor gets its one AccumulateGrad after this point: finish() finds and repairs that
            grads, flat, work = _reduce_bucket(sel, self.mean, async_op=True)
            # what left: the tensor and its version per entry, so that finish() can tell a gradient that was produced or
            # added to AFTER this point (the scheduling property above is the engine's, not a contract)
            self.pending = (sel, grads, flat, work, [g._version for g in grads])

    def _on_early_done(self, module, grad_input, grad_output):
        if _active() and self.pending is None:
            self._launch_early([p for p in self.model.parameters() if p.requires_grad])
        return None

    def finish(self):
        """The rest of average_gradients(model, mean=self.mean)."""
        import sys
        st = sys.modules.get("rslo_amd.streams")
        if st is not None:              # called from inside a backward pass (apex DDP stand-in): the collected weight-gradient
            st.flush_deferred()         # reduces of the issuing stream must be in front of the buckets
        if not _active():
            return
        params = [p for p in self.model.parameters() if p.requires_grad]
        if self.pending is None:        # the hook did not fire on this rank (its loss never reached the early module):
            self._launch_early(params)  # the other ranks sent the early bucket, this one joins it now
        pend, self.pending = self.pending, None
        # gradients of early parameters that changed after the early bucket left (a parameter also used outside the watched
        # boundary, a watch(tensor) that is not a true cut): 1 = produced late (the bucket carried a zeros stand-in: this
        # rank's contribution is still missing from the sum), 2 = ADDED to in place after the copy was taken (the late
        # part cannot be separated any more).  The flag rides the host exchange of the set agreement: all ranks act alike.
        late, flag = [], 0
        if pend is not None:
            for k, (p, g, v) in enumerate(zip(pend[0], pend[1], pend[4])):
                if p.grad is not None and p.grad is not g:
                    late.append(k)
                    flag = max(flag, 1)
                elif p.grad is g and g._version != v:
                    flag = 2
        idx, flag = _agreed_grad_set_and_flag(self.model, params, flag)
        if flag >= 2:
            if pend is not None:
                pend[3].wait()
            raise RuntimeError("OverlappedGradientExchange: a gradient of the early module was accumulated into after the "
                               "early bucket had left (on some rank) -- the watched tensor is not a cut of the graph; use "
                               "average_gradients() or watch the right boundary")
        done = {id(p) for p in pend[0]} if pend is not None else set()
        rest = [params[i] for i in idx if id(params[i]) not in done]
        if rest:
            grads, flat, _ = _reduce_bucket(rest, self.mean)
            _scatter_bucket(rest, grads, flat, self.mean)
        if pend is not None:
            sel, grads, flat, work = pend[:4]
            work.wait()
            if flag == 1:      # some rank holds late gradients: one more flat all-reduce over the same tensors (zeros elsewhere)
                mine = set(late)
                extra = torch.cat([(p.grad if k in mine else torch.zeros_like(p)).reshape(-1) for k, p in enumerate(sel)])
                dist.all_reduce(extra)
                flat.add_(extra)
                for k in late:
                    sel[k].grad = None          # the stand-in takes over with the complete sum
            _scatter_bucket(sel, grads, flat, self.mean)


def broadcast_params(model, src=0):
    """Rank `src` -> all: parameters and buffers, one flattened broadcast per dtype."""
    if not _active():
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


class DistModule(Module):
    """Keeps replicas in sync at construction (broadcast); gradients are reduced explicitly by
    `average_gradients` after backward."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        broadcast_params(self.module)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)


class ParallelWrapper(Module):
    def __init__(self, net, parallel_mode="none"):
        super().__init__()
        assert parallel_mode in ("dist", "data_parallel", "none")
        self.parallel_mode = parallel_mode
        if parallel_mode == "none":
            self.net = net
        elif parallel_mode == "dist":
            self.net = DistModule(net)
        else:
            raise NotImplementedError("single-process DataParallel is not an MI355X configuration: "
                                      "one process per GPU (parallel_mode='dist')")
        self.module = net

    def forward(self, *inputs, **kwargs):
        return self.net(*inputs, **kwargs)


def dist_init(port):
    """SLURM launch (SLURM_PROCID / SLURM_NTASKS / SLURM_NODELIST) -> RCCL process group; returns (rank, world).
    The first host of the node list becomes MASTER_ADDR; unlike the reference no site-specific host-name slicing
    (distributed_utils.py:96-101) is applied."""
    import multiprocessing as mp
    if mp.get_start_method(allow_none=True) != "spawn":
        mp.set_start_method("spawn", force=True)
    rank = int(os.environ["SLURM_PROCID"])
    world = int(os.environ["SLURM_NTASKS"])
    nodes = os.environ["SLURM_NODELIST"]
    if "[" in nodes:                      # "prefix[a-b,c]" -> "prefixa"
        head, rest = nodes.split("[", 1)
        first = rest.replace("]", "").split(",")[0].split("-")[0]
        nodes = head + first
    addr = nodes.split(",")[0]
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())
    os.environ.update(MASTER_PORT=str(port), MASTER_ADDR=addr, WORLD_SIZE=str(world), RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl")
    return dist.get_rank(), dist.get_world_size()


def _world_rank(world_size, rank):
    if world_size is None:
        world_size = dist.get_world_size()
    if rank is None:
        rank = dist.get_rank()
    assert rank < world_size
    return world_size, rank


def _n(dataset):
    return dataset if isinstance(dataset, int) else len(dataset)


class DistributedSequatialSampler(Sampler):
    """Un-shuffled strided shard (evaluation): rank r gets indices r, r + R, r + 2R, ... of the dataset padded by
    wrap-around to a multiple of R (reference :117-175)."""

    def __init__(self, dataset, num_replicas=None, rank=None):
        num_replicas, rank = _world_rank(num_replicas, rank)
        self.dataset, self.num_replicas, self.rank, self.epoch = dataset, num_replicas, rank, 0
        self.num_samples = int(math.ceil(_n(dataset) / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        idx = list(range(_n(self.dataset)))
        idx += idx[:self.total_size - len(idx)]
        return iter(idx[self.rank:self.total_size:self.num_replicas])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class _GivenIteration(Sampler):
    """total_iter * batch_size samples for this rank: every rank builds the same global list (seed 7) and takes the
    contiguous slice [total_size * rank, total_size * (rank + 1)); iteration resumes after `last_iter`."""

    def __init__(self, dataset, total_iter, batch_size, world_size=None, rank=None, last_iter=-1):
        self.world_size, self.rank = _world_rank(world_size, rank)
        self.dataset, self.total_iter, self.batch_size, self.last_iter = dataset, total_iter, batch_size, last_iter
        self.total_size = total_iter * batch_size
        self.call = 0
        self.indices = self.gen_new_list()

    def __iter__(self):
        self.call = 1
        return iter(self.indices[(self.last_iter + 1) * self.batch_size:])

    def __len__(self):          # display only: the resume offset is not subtracted (reference :227-232)
        return self.total_size

    def set_epoch(self, epoch):
        pass

    def _shard(self, arr):
        beg = self.total_size * self.rank
        out = arr[beg:beg + self.total_size]
        assert len(out) == self.total_size
        return out


class DistributedGivenIterationSampler(_GivenIteration):
    """Tile the dataset to the global size, ONE shuffle of the whole list (reference :178-236)."""

    def gen_new_list(self):
        np.random.seed(7)
        all_size = self.total_size * self.world_size
        base = np.arange(_n(self.dataset))[:all_size]
        reps = (all_size - 1) // base.shape[0] + 1
        idx = np.tile(base, reps)[:all_size]
        np.random.shuffle(idx)
        return self._shard(idx)


class DistributedGivenIterationSamplerEpoch(_GivenIteration):
    """Epoch-wise: a fresh permutation of the dataset per epoch, concatenated; yields (index, seed) where seed is
    the position in the global list (the dataset uses it to seed its per-sample augmentation).  review_cycle = c > 0
    (1/c integral) replays every block of c * len(dataset) samples twice in a row (reference :238-314)."""

    def __init__(self, dataset, total_iter, batch_size, world_size=None, rank=None, last_iter=-1, review_cycle=-1):
        self.review_cycle = review_cycle
        super().__init__(dataset, total_iter, batch_size, world_size, rank, last_iter)

    def gen_new_list(self):
        np.random.seed(7)
        all_size = self.total_size * self.world_size
        base = np.arange(_n(self.dataset))[:all_size]
        reps = (all_size - 1) // base.shape[0] + 1
        idx = np.concatenate([np.random.permutation(base) for _ in range(reps)])
        seeds = np.arange(idx.size)
        if self.review_cycle > 0:
            assert (1 / self.review_cycle) % 1 == 0
            block = int(self.review_cycle * _n(self.dataset))
            rows = len(idx) // block
            idx = np.tile(idx[:rows * block].reshape(rows, block), (1, 2)).reshape(-1)
            seeds = np.tile(seeds[:rows * block].reshape(rows, block), (1, 2)).reshape(-1)
        idx, seeds = self._shard(idx[:all_size]), self._shard(seeds[:all_size])
        return list(zip(list(idx), list(seeds)))
