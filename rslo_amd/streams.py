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
_state = {}          # device -> {"side", "cur": stream of the backward nodes, "pending", "targets": ids, "keep": inputs}
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
                if st["pending"] and not st.get("queued"):
                    st["queued"] = True
                    torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev: join(d))
        return False


def join(device=None):
    """Make the streams that issued leaf work wait for it (no-op when nothing is pending)."""
    for dev, st in _state.items():
        if st["pending"] and (device is None or dev == device):
            from rslo_amd import streamprobe
            streamprobe.wait("leaf_wgrad_stream", st["cur"], lambda: st["cur"].wait_stream(st["side"]))
            st["pending"] = False
            st["queued"] = False
            st["targets"].clear()
            st["keep"].clear()


def leaf(fn, inputs, params=None):
    """fn() -> tensor or tuple of tensors (or None entries); inputs: the CUDA tensors fn reads; params: the parameters
    whose gradients fn's results are (see the module docstring)."""
    dev = inputs[0].device
    if not (ENABLED and dev.type == "cuda") or params is None:
        return fn()
    cur = torch.cuda.current_stream(dev)
    st = _state.get(dev)
    if st is None:
        st = _state[dev] = {"side": torch.cuda.Stream(dev), "cur": cur, "pending": False, "targets": set(), "keep": []}
    # (a non-leaf "parameter", e.g. a cast copy, hands the result to another backward node on the issuing stream)
    if any((not p.is_leaf) or p.grad is not None or id(p) in st["targets"] for p in params):
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
    st["targets"].update(id(p) for p in params)
    if not st["pending"]:
        st["pending"], st["cur"] = True, cur
        if _hold[0] == 0:
            st["queued"] = True
            torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev: join(d))
    return out
