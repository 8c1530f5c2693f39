"""The BEV head's training pass with its FORWARD replayed from a hipGraph (default), optionally its backward as well.

The head (reference: rslo/models/odom_pred.py:136-300, UNOdomPredEncDecSVDTempMask) works on maps of a fixed size: its 13
residual blocks, 3 deblocks, pyramid / confidence branches and the element-wise tail are the same ~170 launches forward and
~180 backward in every step, most of them on 24x44 / 12x22 maps where a kernel lasts 5-10 us -- less than the interpreter
needs to issue it.  Issuing one step costs the two host threads 9.3-11 ms (forward 4-5 on the main thread, backward 5-6 on the
autograd thread), the GPU needs ~11 ms: on the fast hosts of the pool the GPU is the bound, on the slower ones the host is.

Mode "fwd" (default): the head's forward over a STATIC input map is captured once (torch.cuda.graph around the same modules:
every capi launch goes to the capturing stream -- the caller's own stream whenever that is not the legacy default one).  The
capture also builds the forward's autograd graph over static buffers; it is RETAINED, and every later step is: copy the
encoder's BEV map into the static input, replay (one launch instead of ~170), and in backward run torch.autograd.grad over the
retained graph -- the backward kernels issued launch by launch as ever, weight gradients on the leaf stream (rslo_amd/
streams.py, joined at the end of the ENCLOSING pass).  Parameters and the static input enter the capture as storage aliases
with their own version counters, so the optimizer's in-place updates and the per-step input copy do not trip the retained
graph's version checks.  Mode "full" (RSLO_HEAD_GRAPH=1): the backward is captured too (torch.autograd.grad inside a second
capture) and replayed; static parameter gradients are handed to `p.grad`.
Same kernels, same order per stream, same bits as the eager pass in both modes (tests/test_gpu_model.py).

Measured, round 5 (profiles/NOTES.md), ms per step eager | fwd | full: C3 (fp32) on a fast host 11.12 | 11.20 | 13.5 (11.9 with
a linear backward graph), on a slower host 11.55-11.91 | 10.98-11.21; C4 (bf16 operands, less GPU work) 10.9-13.0 | 9.2-9.4.
"fwd" never loses more than the noise and wins wherever the host is the bound; "full" loses on ROCm 7.2 (a multi-branch
hipGraph runs behind cross-queue barriers and takes 0.8 ms to launch) and stays an option.

Not captured: the sparse encoder and the losses (row counts change with every scan), the optimizer (already 3 launches), the
weight-operand split (runs beside the encoder, rslo/layers/hip_conv2d.py presplit_early) and the covariance branch.
Multi-rank SyncBatchNorm (round 6): the statistics exchanges of the node's peer comm are launched inside the capture with
numbers relative to a device word that is set in front of every replay (rslo_amd/peer.py capture_begin / replay_prepare), so a
rank of an N > 1 job replays the same graph the one-rank step does; exchanges that are RCCL collectives (several hosts,
RSLO_SYNCBN_EXCHANGE=rccl) keep the head eager.
The eager pass stays the path for eval / no-grad calls, other dtypes, the first WARM_CALLS calls of a shape (lazy initialisation must not be captured), a second training forward
issued while the first one's backward is still outstanding (the static activations belong to the first), and any head whose
capture fails (warned once).  RSLO_HEAD_GRAPH=0 turns the graphs off.
"""
import os
import warnings
import weakref

import torch

# "0" (default) | "fwd": the forward replayed, the backward issued launch by launch over the retained autograd graph of the capture
# | "1" / "full": forward and backward replayed
MODE = {"1": "full", "full": "full", "fwd": "fwd"}.get(os.environ.get("RSLO_HEAD_GRAPH", "fwd"))
ENABLED = MODE is not None
DIRECT_INPUT = os.environ.get("RSLO_HEAD_GRAPH_INPUT", "direct") != "copy"      # static_input(): dense() writes into the graph's input
WARM_CALLS = 1      # the first call of a shape runs eagerly (lazy initialisation: workspaces, operand plans, fused-BN state)
_STATE = weakref.WeakKeyDictionary()      # head module -> _State (graphs neither deep-copy nor pickle: kept off the module)


def _flatten(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
        return ("t", len(out) - 1)
    if isinstance(obj, dict):
        return ("d", [(k, _flatten(v, out)) for k, v in obj.items()])
    if isinstance(obj, (list, tuple)):
        return ("l" if isinstance(obj, list) else "u", [_flatten(v, out) for v in obj])
    return ("c", obj)


def _unflatten(spec, ts):
    kind, val = spec
    if kind == "t":
        return ts[val]
    if kind == "d":
        return {k: _unflatten(v, ts) for k, v in val}
    if kind == "l":
        return [_unflatten(v, ts) for v in val]
    if kind == "u":
        return tuple(_unflatten(v, ts) for v in val)
    return val


class _State:
    def __init__(self):
        self.key, self.calls, self.graph, self.failed, self.mods, self.hooks, self.params = None, 0, None, False, None, None, None
        self.parked = None          # (key, graphs) of the shape used before the current one


def _storage_alias(t):
    """A tensor over the same memory as `t` with its OWN autograd version counter (detach() shares the counter): writes through
    it, and the optimizer's in-place updates of the parameter it aliases, are invisible to the version checks of a retained
    autograd graph that saved the other tensor -- the retained graph reads whatever the memory holds when it runs, which is
    exactly what a replayed forward needs."""
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())


class _HeadGraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hg, base):
        ctx.hg = hg
        ctx.set_materialize_grads(False)     # outputs the loss does not differentiate arrive as None, not as zero-filled maps
        if base.data_ptr() != hg.static_in.data_ptr():
            hg.input_writer.copy_(base)
        if hg.n_fwd_exchanges:       # multi-rank SyncBN inside the capture: this replay's exchange numbers (rslo_amd/peer.py)
            hg.comm.replay_prepare(hg.n_fwd_exchanges)
        hg.replay(hg.g_fwd, "head_graph_forward")
        return tuple(hg.flat[i].detach() for i in hg.req)

    @staticmethod
    def backward(ctx, *gouts):
        hg = ctx.hg
        hg.awaiting = False
        if hg.g_bwd is None:
            # mode "fwd": the backward pass is issued launch by launch (weight gradients on the leaf stream, as ever) over
            # the autograd graph the captured forward built -- retained, its saved tensors are the static buffers the replay
            # has just refilled
            from rslo_amd import streams
            # only the outputs the loss differentiates are roots of the nested pass: a zero stand-in for an unused one (the
            # confidence maps returned for logging) is ADDED to the real gradient of the same tensor by the engine
            live = [k for k, g in enumerate(gouts) if g is not None]
            if not live:
                return None, None
            roots, gs = [hg.raw_req[k] for k in live], [gouts[k] for k in live]
            if hg.on_callers_stream:
                with streams.join_in_enclosing_pass():
                    grads = torch.autograd.grad(roots, [hg.static_in] + hg.aliases, gs, retain_graph=True, allow_unused=True)
            else:       # captured on a side stream (the caller sits on the legacy default stream): the nodes run there and the
                prev, streams.ENABLED = streams.ENABLED, False      # engine orders that stream against the caller's at the end of
                try:                                                # the nested pass -- which only covers work issued ON it
                    grads = torch.autograd.grad(roots, [hg.static_in] + hg.aliases, gs, retain_graph=True, allow_unused=True)
                finally:
                    streams.ENABLED = prev
            # The alias leaves never carry a gradient, so streams.leaf() always took the side stream for them; an ADD into a
            # gradient that is already there (accumulation over several backward() calls, zero_grad(set_to_none=False)) reads
            # `g` on the issuing stream: the leaf stream is joined first, exactly as streams.leaf() does on the eager pass
            if any(g is not None and p.grad is not None for p, g in zip(hg.params, grads[1:])):
                streams.join(hg.device)
            for p, g in zip(hg.params, grads[1:]):
                if g is not None:
                    p.grad = g if p.grad is None else p.grad + g
            return None, grads[0]
        for k, g in enumerate(gouts):
            if g is None:
                if not hg.gout_zero[k]:
                    hg.gouts[k].zero_()
                    hg.gout_zero[k] = True
            else:
                hg.gouts[k].copy_(g)
                hg.gout_zero[k] = False
        # a caller that did not reset the gradients expects a SUM: the replay overwrites the static buffers, so a static
        # buffer still installed as p.grad is copied out first (the shipped loop resets to None every step: no copy)
        for p, sg in hg.pgrads:
            if p.grad is sg:
                p.grad = sg.clone()
        if hg.n_bwd_exchanges:
            hg.comm.replay_prepare(hg.n_bwd_exchanges)
        hg.replay(hg.g_bwd, "head_graph_backward")
        for p, sg in hg.pgrads:
            if p.grad is None:
                p.grad = sg
            else:
                p.grad = p.grad + sg
        return None, hg.grad_in


class HeadGraph:
    def __init__(self, head, base, T, mode):
        from apex import parallel as apex_parallel
        self.T, self.mode = T, mode
        dev = base.device
        named = [(n, p) for n, p in head.named_parameters() if p.requires_grad]
        self.params = params = [p for _, p in named]
        # The capture runs over ALIASES of the parameters (same storage, fresh leaves): a parameter's AccumulateGrad node may be
        # alive from an earlier step and belongs to the stream of that step -- the engine would then order the capturing stream
        # against it inside the capture (on the legacy default stream that ends the capture with a fault).  An alias gets its node
        # here, on the capturing stream; it also has no gradient yet, which is what the leaf stream (rslo_amd/streams.py) wants to
        # see before it takes a weight gradient.  The split operands the layers look up on the parameter object ride along.
        alias = {}
        for n, p in named:
            a = _storage_alias(p).requires_grad_(True)
            sp = getattr(p, "_hip_split", None)
            if sp is not None:
                a._hip_split = (sp[0], sp[1], a._version, a.data_ptr()) if (sp[2] == p._version and sp[3] == p.data_ptr()) else None
            alias[n] = a
        self.aliases = [alias[n] for n, _ in named]
        self.static_in = torch.empty_like(base).requires_grad_(True)
        self.input_writer = _storage_alias(self.static_in)
        self.input_writer.copy_(base)
        pool = torch.cuda.graph_pool_handle()
        self.g_fwd = torch.cuda.CUDAGraph()
        self.g_bwd = torch.cuda.CUDAGraph() if mode == "full" else None
        isolate = getattr(apex_parallel, "isolated_batch_counts", None)
        # mode "fwd": the backward nodes run on the stream their forward was captured on -- capture on the caller's stream
        # (the training stream with its priority) whenever that is not the legacy default stream
        cur = torch.cuda.current_stream(dev)
        on = {"stream": cur} if (mode == "fwd" and cur.cuda_stream != 0) else {}
        self.on_callers_stream = bool(on)
        # Multi-rank SyncBatchNorm inside the capture: the layers' statistics exchanges (same-stream kernels of the node's peer
        # comm) are launched with numbers RELATIVE to a device word that replay_prepare() sets in front of every replay, so
        # their launch arguments do not change from replay to replay (csrc/peer.hip rslo_peer_capture_begin).
        self.comm = _peer_comm() if _multi_rank() else None
        self.n_fwd_exchanges = self.n_bwd_exchanges = 0
        head.__dict__["_in_graph_capture"] = True
        try:
            with (isolate() if isolate is not None else _null()):
                if self.comm is not None:
                    self.comm.capture_begin()
                try:
                    with torch.cuda.graph(self.g_fwd, pool=pool, capture_error_mode="thread_local", **on):
                        frames = list(self.static_in.split(base.shape[1] // T, dim=1))
                        for f in frames:
                            f._pair_base = self.static_in
                        with torch.enable_grad():
                            out = torch.func.functional_call(head, alias, (frames,))
                finally:
                    if self.comm is not None:
                        self.n_fwd_exchanges = self.comm.capture_end()
            self.flat = []
            self.spec = _flatten(out, self.flat)
            self.req = [i for i, t in enumerate(self.flat) if t.requires_grad]
            self.raw_req = [self.flat[i] for i in self.req]
            if mode == "full":
                self.gouts = [torch.zeros_like(self.flat[i]) for i in self.req]
                self.gout_zero = [True] * len(self.req)
                if self.comm is not None:
                    self.comm.capture_begin()
                try:
                    with torch.cuda.graph(self.g_bwd, pool=pool, capture_error_mode="thread_local"):
                        grads = torch.autograd.grad(self.raw_req, [self.static_in] + self.aliases, self.gouts, allow_unused=True)
                finally:
                    if self.comm is not None:
                        self.n_bwd_exchanges = self.comm.capture_end()
                self.grad_in = grads[0]
                self.pgrads = [(p, g) for p, g in zip(params, grads[1:]) if g is not None]
                self.raw_req = None
        finally:
            head.__dict__.pop("_in_graph_capture", None)
        self.flat = [t.detach() for t in self.flat]
        self.device = dev
        self.awaiting, self._node = False, None

    @staticmethod
    def replay(g, name):
        from rslo_amd import streamprobe
        if streamprobe.enabled():
            with streamprobe.span("op", name):
                g.replay()
        else:
            g.replay()

    def busy(self):
        """True while the outputs of the latest replayed forward can still be back-propagated (their node is alive and no
        backward has run): a second forward would overwrite the static activations that backward needs."""
        return self.awaiting and self._node is not None and self._node() is not None

    def forward(self, base):
        outs = _HeadGraphFn.apply(self, base)
        self.awaiting = True
        self._node = weakref.ref(outs[0].grad_fn) if outs and outs[0].grad_fn is not None else None
        ts = list(self.flat)
        for i, o in zip(self.req, outs):
            ts[i] = o
        return _unflatten(self.spec, ts)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _multi_rank():
    import torch.distributed as dist
    if os.environ.get("RSLO_FORCE_SYNCBN_PATH", "0") == "1":
        return dist.is_available() and dist.is_initialized()
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _peer_comm():
    """The node's peer comm if it EXISTS already (created by the first eager forward: a collective call, never made here), else
    None.  With it every SyncBN exchange of the head is a same-stream kernel that can go into the capture; without it (several
    hosts, RSLO_SYNCBN_EXCHANGE=rccl, an explicit process group) the exchanges are RCCL collectives and the head stays eager."""
    import sys
    import torch.distributed as dist
    pr = sys.modules.get("rslo_amd.peer")
    if pr is None:
        return None
    ent = pr._COMMS.get(id(dist.group.WORLD))
    if ent is None or ent[0] is not dist.group.WORLD or not ent[1]:
        return None
    return ent[1]


def wanted(head, base, T):
    """True when this call of the head goes through run(): a training pass over one fp32 CUDA pair map by a head that has the
    hook points, on one rank.  (The first WARM_CALLS calls of a shape still come back from run() as None = eager.)"""
    if not ENABLED or base is None or not (head.training and torch.is_grad_enabled()):
        return False
    if not (base.is_cuda and base.dtype == torch.float32 and base.requires_grad and base.dim() == 4 and T == 2):
        return False
    if not getattr(head, "graph_capturable", False):      # (T == 2: the cycle-constraint pairing is the identity)
        return False
    if getattr(head, "use_svd", False) or getattr(head, "track_masks", False):      # a library SVD reads back to the host
        return False
    if _multi_rank() and _STATE.get(head) is not None and _STATE[head].calls >= WARM_CALLS and _peer_comm() is None:
        return False      # (the warm-up call of a shape is eager anyway and is what creates the comm)
    st = _STATE.get(head)
    return st is None or not st.failed


def static_input(head, T, batch):
    """The buffer the NEXT training forward of `head` would copy its input map into -- the static input of the current graph --
    or None: no graph yet, graphs off, eval / no-grad, a forward still waiting for its backward (its static activations are
    in use).  The encoder's dense() writes the BEV map there directly (70 MB less copied per step); run() sees the same address
    and skips its copy.  A caller that ends up with a map of another shape simply does not use the buffer."""
    if not ENABLED or not DIRECT_INPUT or not (head.training and torch.is_grad_enabled()) or T != 2:
        return None
    st = _STATE.get(head)
    if st is None or st.graph is None or st.failed or st.graph.busy() or st.key is None or st.key[3] != MODE:
        return None
    if st.key[0][0] != batch or st.key[2] != T:
        return None
    if any(st.hooks) or any("forward" in d for d in st.mods):
        return None
    from rslo_amd import precision
    if st.key[4] != precision.low_precision() or st.key[5] != tuple([p.data_ptr() for p in st.params]):
        return None
    return st.graph.input_writer


def run(head, base, T):
    """The head's training forward over `base` ([B, T*C, H, W], the frames of a pair side by side) -> its prediction dict, or
    None: the caller runs the eager pass (warm-up calls of a shape, or a capture that failed -- warned once)."""
    st = _STATE.get(head)
    if st is None:
        st = _STATE[head] = _State()
    # the graphs hold the addresses of the parameters (and of everything else they touch): a head whose storage moved is a new one
    from rslo_amd import precision
    if st.mods is None:      # walked once per head (Module.parameters() / .modules() cost ~1 ms per call on this head)
        mods = list(head.modules())
        st.mods = [m.__dict__ for m in mods]
        st.hooks = [d for m in mods for d in (m._forward_hooks, m._forward_pre_hooks, m._backward_hooks, m._backward_pre_hooks)]
        st.params = [p for p in head.parameters() if p.requires_grad]
    key = (tuple(base.shape), base.device, T, MODE, precision.low_precision(), tuple([p.data_ptr() for p in st.params]))
    if st.key != key:
        # a second shape (the short last batch of an epoch) gets its own graphs; the previous one is parked and comes back
        # without a new capture -- two shapes are kept, an older one is dropped with its memory pool
        prev = (st.key, st.graph) if st.graph is not None else None
        if st.parked is not None and st.parked[0] == key:
            st.key, st.calls, st.graph, st.parked = key, WARM_CALLS, st.parked[1], prev
        else:
            st.key, st.calls, st.graph = key, 0, None
            if prev is not None:
                st.parked = prev
    # a replay runs the captured kernels, not the modules: hooks on a submodule and instance-level `forward` overrides (tests,
    # debugging) would silently not run -- such a head takes the eager pass
    if any(st.hooks) or any("forward" in d for d in st.mods):
        return None
    if st.graph is None and st.calls < WARM_CALLS:
        st.calls += 1
        return None
    if st.graph is not None and st.graph.busy():
        # two training forwards before a backward (losses of several batches summed): the static activations of the first
        # are still needed -- this call takes the eager pass, whose activations are its own
        return None
    # the split weight operands of this step (hip_conv2d.presplit_early, beside the encoder): waited for / made OUTSIDE the
    # graphs, which only read them
    from rslo.layers import hip_conv2d
    ev = head.__dict__.pop("_presplit_event", None)
    if ev is not None:
        from rslo_amd import streamprobe
        cur = torch.cuda.current_stream(base.device)
        streamprobe.wait("head_weight_presplit", cur, lambda: cur.wait_event(ev))
    else:
        hip_conv2d.presplit(head)
    if st.graph is None:
        try:
            st.graph = HeadGraph(head, base, T, MODE or "full")
        except Exception as e:      # the eager pass runs the same kernels: slower issue, same results
            st.failed = True
            from rslo_amd import streams
            streams.join()
            warnings.warn("rslo_amd.headgraph: capture of the BEV head failed (%s: %s); the head stays on the eager pass"
                          % (type(e).__name__, e))
            return None
    return st.graph.forward(base)
