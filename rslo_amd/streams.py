"""Leaf work of the backward pass on a second HIP stream.

A weight gradient is a LEAF of backward: nothing downstream in the pass reads it, while the data-gradient chain it hangs
off is a sequence of dependent launches that leave most CUs idle on the small maps / levels.  `leaf(fn, inputs)` runs
`fn` on a per-device side stream ordered behind the issuing stream's current position, and queues ONE engine callback
that makes the issuing stream wait for the side stream at the end of the backward pass (`join()` does the same on
request, e.g. before a gradient bucket leaves mid-backward).  RSLO_WGRAD_STREAM=0 runs everything on the issuing stream.
Safety at the node boundary (the autograd engine knows nothing about the side stream):
  * `params` names the parameters the results become gradients of.  AccumulateGrad runs on the ISSUING stream: when it
    only takes the tensor (`p.grad is None`, the shipped loop: `zero_grad()` to None every step) nothing reads the result
    before the end-of-pass join.  When it would ADD -- `p.grad` already set (gradient accumulation over several
    backward() calls, `zero_grad(set_to_none=False)`) or a second result for the same parameter in one pass (a weight
    shared by two layers) -- the side stream is joined first and `fn` runs on the issuing stream.  A call without
    `params` cannot be checked and runs on the issuing stream.
  * the inputs are kept referenced until the join: the engine adds a later gradient IN PLACE into a buffered one only
    when it is its sole owner (`InputBuffer::add`), so a `dy` the side stream is still reading is never written.
Used for the dense weight gradients of the BEV head (12.68 -> 12.29 ms per step); the sparse weight gradients were tried
as well and lost (12.44-12.69 ms: their launches fill the GPU on levels 0-2, so they only take CUs from the data-gradient
chain they would run beside).
"""
import os

import torch

ENABLED = os.environ.get("RSLO_WGRAD_STREAM", "1") != "0"
# The step tolerates FOUR streams with work on them (training, structure plan, covariance branch, leaf work): a fifth one -- a
# single tiny kernel per step on any further stream, which is what RCCL's own stream is in a data-parallel job -- ran the C3
# step at HALF speed (21.3 vs 10.6 ms, profiles/r06_fifth_stream.txt; not the hardware-queue count, not a priority matter; it
# does not happen without the leaf stream; two further streams break the shared form as well: 23.4 ms).
# "auto" (default): in a process with an initialised process group the dense weight gradients are issued on the covariance
# branch's stream -- three streams of ours, the fourth slot left to the collective library; "1": always; "0": never.  Sharing
# costs the step nothing at one rank (10.52-10.59 vs 10.54-10.62 ms) but the side work then reaches further into the encoder's
# backward: the sparse kernels of the training stream measure 98 instead of 91 us per launch inside the step, which is why a
# single process keeps the two streams apart.
SHARE_SIDE = os.environ.get("RSLO_SHARE_SIDE_STREAM", "auto")
_shared_side = {}


def sharing():
    if SHARE_SIDE in ("0", "1"):
        return SHARE_SIDE == "1"
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def side_stream(dev):
    """The one side stream of `dev` that the covariance branch and the leaf work share when sharing() is on."""
    dev = torch.device(dev)
    s = _shared_side.get(dev)
    if s is None:
        s = _shared_side[dev] = torch.cuda.Stream(dev)
    return s

# "1" (default): the second stage of every weight gradient of a backward pass (slab / chunk partials -> gradient, a 2-10 us
# launch per layer: 45 dense 3x3, 5 dense 1x1, 20 sparse) is collected and run as ONE launch per stream at the end of the pass
# (rslo_amd.capi.ReduceSink, csrc/wgrad_reduce.hip): the dense ones on the leaf stream in front of its join, the sparse ones on
# the issuing stream in an end-of-pass callback.  Same block bodies, same bits.  "0": a reduce launch per layer, as before.
DEFER_REDUCES = os.environ.get("RSLO_DEFER_WGRAD_REDUCE", "8") != "0"
_DEFER_WHICH = os.environ.get("RSLO_DEFER_WGRAD_REDUCE", "8")      # (debugging: "dense" / "sparse" = only that family)
# a sink is flushed when it holds this many layers (and at the end of the pass): the partials of ~8 layers (~100 MB) are still in
# the 256 MB Infinity Cache when their reduce reads them, the partials of a whole pass (~900 MB) are not ("1": only at the end)
FLUSH_EVERY = int(_DEFER_WHICH) if _DEFER_WHICH.isdigit() and int(_DEFER_WHICH) >= 2 else (1 << 30)
_state = {}          # device -> {"side", "cur": stream of the backward nodes, "pending", "targets": {pass -> ids}, "keep": inputs}
_hold = [0]          # > 0: inside join_in_enclosing_pass()


class join_in_enclosing_pass:
    """A backward pass run from INSIDE a node of another one (rslo_amd/headgraph.py: torch.autograd.grad over a retained graph):
    leaf work issued in it is joined at the end of the ENCLOSING pass, not of the nested one -- the trailing weight gradients
    keep running beside whatever the enclosing pass issues next.  The callback is queued at exit, in the enclosing pass."""

    def __enter__(self):
        _hold[0] += 1
        return self

    def __exit__(self, *exc):
        _hold[0] -= 1
        if _hold[0] == 0:
            for dev, st in _state.items():
                for pid in st.pop("nested", ()):          # the nested passes are over: nothing of theirs can arrive any more
                    st["targets"].pop(pid, None)
                if st["pending"] and not st.get("queued"):
                    st["queued"] = True
                    torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev, p=_pass_id(): join(d, p))
        return False


def _flush_leaf_sink(st):
    sink = st.get("sink")
    if sink is not None and (sink.pending() or sink.keep):
        cur = torch.cuda.current_stream(st["side"].device)
        torch.cuda.set_stream(st["side"])
        try:
            sink.flush()
        finally:
            torch.cuda.set_stream(cur)


def _pass_id():
    """The running backward pass (graph task), -1 outside one."""
    try:
        return torch._C._current_graph_task_id()
    except AttributeError:
        return -1


def _targets(st):
    """The parameters that already got a result in THIS backward pass.  Kept until the pass is over -- not until the next join: the
    engine adds a parameter's second contribution to the buffered first one as soon as it arrives (InputBuffer::add), whatever
    joins happened in between."""
    by_pass = st["targets"]
    pid = _pass_id()
    t = by_pass.get(pid)
    if t is None:
        if len(by_pass) > 8:          # passes whose end-of-pass callback never ran (an exception inside backward)
            by_pass.clear()
        t = by_pass[pid] = set()
    return t


def join(device=None, done_pass=None):
    """Make the streams that issued leaf work wait for it (no-op when nothing is pending).  The collected weight-gradient
    reduces of the leaf stream are launched in front of the wait.  done_pass: the backward pass that is over (its end-of-pass
    callback)."""
    for dev, st in _state.items():
        if device is None or dev == device:
            if st["pending"]:
                _flush_leaf_sink(st)
                from rslo_amd import streamprobe
                streamprobe.wait("leaf_wgrad_stream", st["cur"], lambda: st["cur"].wait_stream(st["side"]))
                st["pending"] = False
                st["queued"] = False
                st["keep"].clear()
            if done_pass is not None:
                st["targets"].pop(done_pass, None)
                st.get("cleanup", set()).discard(done_pass)


def leaf(fn, inputs, params=None):
    """fn() -> tensor or tuple of tensors (or None entries); inputs: the CUDA tensors fn reads; params: the parameters
    whose gradients fn's results are (see the module docstring)."""
    dev = inputs[0].device
    if not (ENABLED and dev.type == "cuda") or params is None:
        return fn()
    cur = torch.cuda.current_stream(dev)
    st = _state.get(dev)
    if st is None:
        st = _state[dev] = {"side": side_stream(dev) if sharing() else torch.cuda.Stream(dev), "cur": cur, "pending": False, "targets": {}, "keep": []}
    # (a non-leaf "parameter", e.g. a cast copy, hands the result to another backward node on the issuing stream)
    targets = _targets(st)
    pid = _pass_id()
    if _hold[0] > 0:
        st.setdefault("nested", set()).add(pid)
    elif pid not in st.setdefault("cleanup", set()):      # the pass's entry goes when the pass is over, joined or not
        st["cleanup"].add(pid)
        torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev, p=pid: join(d, p))
    if any((not p.is_leaf) or p.grad is not None or id(p) in targets for p in params):
        targets.update(id(p) for p in params)
        join(dev)              # everything issued so far is ordered in front of the accumulation that follows
        return fn()
    side = st["side"]
    # one event per device, recorded anew for every call (a wait takes the state of the event at the time of the call); inside a
    # stream capture every fork gets its own
    ev = st.get("event")
    if ev is None or torch.cuda.is_current_stream_capturing():
        ev = torch.cuda.Event()
        if not torch.cuda.is_current_stream_capturing():
            st["event"] = ev
    ev.record(cur)
    side.wait_event(ev)
    torch.cuda.set_stream(side)
    try:
        if DEFER_REDUCES and _DEFER_WHICH != "sparse":
            sink = st.get("sink")
            if sink is None:
                from rslo_amd import capi
                sink = st["sink"] = capi.ReduceSink()
            with sink.collect():
                out = fn()
            if sink.pending() >= FLUSH_EVERY:          # (on the side stream, behind the kernels that wrote the partials)
                sink.flush()
        else:
            out = fn()
    finally:
        torch.cuda.set_stream(cur)
    # the inputs stay referenced until the join, after which the issuing stream is ordered behind every read of the side
    # stream: no block of theirs can be handed out again before that (no record_stream needed for them)
    for t in inputs:
        if t is not None:
            st["keep"].append(t)
    for t in (out if isinstance(out, (tuple, list)) else (out,)):
        if t is not None:
            t.record_stream(cur)
    targets.update(id(p) for p in params)
    if not st["pending"]:
        st["pending"], st["cur"] = True, cur
        if _hold[0] == 0:
            st["queued"] = True      # (the pass's clean-up callback, queued above, is the join)
    return out


# ---- weight-gradient reduces of work that stays on the ISSUING stream (the sparse encoder's layers) ------------------------------
_train = {}          # device -> {"sinks": {stream id -> (stream, sink)}, "queued": passes with a callback, "targets": {pass -> data_ptrs}}


def in_backward_pass():
    return _pass_id() != -1


def _flush_sink_on(stream, sink):
    """A sink's reduces run on the stream that wrote its partials (autograd runs a layer's backward on the stream of its forward:
    the covariance branch has its own); the calling stream then waits for it -- whoever flushes is about to read gradients."""
    if not (sink.pending() or sink.keep):
        return
    cur = torch.cuda.current_stream(stream.device)
    if cur.cuda_stream == stream.cuda_stream:
        sink.flush()
        return
    torch.cuda.set_stream(stream)
    try:
        sink.flush()
    finally:
        torch.cuda.set_stream(cur)
    cur.wait_stream(stream)


def deferred_reduce(params):
    """-> a context (capi.ReduceSink.collect()) inside which a weight-gradient entry point leaves its second stage to ONE launch
    per ~8 layers / at the end of the running backward pass, or None when the results may be READ before that: not inside a
    pass, a parameter that already holds a gradient (accumulation: AccumulateGrad adds at once) or got one earlier in this pass
    (shared weight), a non-leaf "parameter", a stream capture.  params: the parameters the results become gradients of.
    One sink per STREAM: the reduce of a layer has to run behind the kernel that wrote its partials."""
    if not DEFER_REDUCES or _DEFER_WHICH == "dense" or not params or params[0].device.type != "cuda" or not in_backward_pass():
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    dev = params[0].device
    st = _train.get(dev)
    if st is None:
        st = _train[dev] = {"sinks": {}, "queued": set(), "targets": {}}
    cur = torch.cuda.current_stream(dev)
    ent = st["sinks"].get(cur.cuda_stream)
    if ent is None:
        from rslo_amd import capi
        ent = st["sinks"][cur.cuda_stream] = (cur, capi.ReduceSink())
    sink = ent[1]
    pid = _pass_id()
    targets = _targets(st)
    seen = any((not p.is_leaf) or p.grad is not None or p.data_ptr() in targets for p in params)
    targets.update(p.data_ptr() for p in params)
    if pid not in st["queued"]:
        st["queued"].add(pid)
        torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev, p=pid: flush_deferred(d, p))
    if sink.pending() >= FLUSH_EVERY:      # the layers collected so far on this stream, while their partials are cache-resident
        sink.flush()
    if seen:
        # a second contribution to a gradient in one pass (two forwards before one backward, a shared weight), or an
        # accumulation: the engine ADDS at once -- everything collected so far is made real, this result is computed in place
        flush_deferred(dev)
        return None
    return sink.collect()


def flush_deferred(device=None, done_pass=None):
    """Launch the collected reduces now, each sink on its own stream, the current stream waiting for the others.  Called at the
    end of the pass (done_pass: that pass), and by anything that reads gradients inside it (the gradient exchange of the apex
    DDP stand-in)."""
    for dev, st in _train.items():
        if device is None or dev == device:
            if done_pass is not None:
                st["queued"].discard(done_pass)
                st["targets"].pop(done_pass, None)
            with torch.cuda.device(dev):
                for stream, sink in list(st["sinks"].values()):
                    _flush_sink_on(stream, sink)
