"""apex.amp surface used by the reference (train_hdf5.py:456-465,663; the `@amp.float_function` islands listed in
SURVEY.md App-B 26), on MI355X terms.

  initialize(models, optimizers, opt_level)   "O0" (the reference default): fp32, nothing changes.
                                              "O1" / "O2": mixed precision = rslo_amd.precision bf16 mode: convolution
                                              operands bf16, fp32 accumulation, fp32 master weights in the SAME parameter
                                              tensors (so the optimizer and the checkpoints are unchanged).  "O3" (pure
                                              half) is refused.
  scale_loss(loss, optimizers)                context manager of the training loop.  bf16 keeps the fp32 exponent range,
                                              so the loss scale is the constant 1: yields `loss` itself; the unscale /
                                              overflow-skip logic of fp16 training has nothing to do.
  float_function / register_float_function    fp32 islands: floating tensor arguments that are not fp32 are promoted, and
                                              torch autocast (if someone enabled it) is switched off inside.  As in apex
                                              O1, convolutions called inside an island still take low-precision operands;
                                              everything else in there computes in fp32 because activations are stored
                                              in fp32.
  master_params(optimizer)                    the parameters themselves.
  state_dict / load_state_dict                {"opt_level", "loss_scale"} for checkpoint round trips.
"""
import contextlib
import functools

import torch

from rslo_amd import precision

_state = {"opt_level": "O0", "loss_scale": 1.0}


def _promote(x):
    if isinstance(x, torch.Tensor) and x.is_floating_point() and x.dtype in (torch.float16, torch.bfloat16):
        return x.float()
    if isinstance(x, (list, tuple)):
        return type(x)(_promote(v) for v in x)
    if isinstance(x, dict):
        return {k: _promote(v) for k, v in x.items()}
    return x


def float_function(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if precision.low_precision() is None and not torch.is_autocast_enabled():
            return fn(*args, **kwargs)
        with torch.autocast(device_type="cuda", enabled=False):
            return fn(*[_promote(a) for a in args], **{k: _promote(v) for k, v in kwargs.items()})

    return wrapper


half_function = promote_function = lambda fn: fn


def register_float_function(module, name):
    setattr(module, name, float_function(getattr(module, name)))


register_half_function = register_promote_function = lambda module, name: None


def initialize(models, optimizers=None, opt_level="O0", enabled=True, loss_scale=None, **kwargs):
    if opt_level not in ("O0", "O1", "O2", "O3"):
        raise ValueError("unknown opt_level %r" % (opt_level,))
    if opt_level == "O3":
        raise NotImplementedError("opt_level O3 (pure half precision) is not a mode of the RSLO path on MI355X")
    if not enabled:
        opt_level = "O0"
    _state["opt_level"] = opt_level
    _state["loss_scale"] = 1.0
    precision.set_low_precision(None if opt_level == "O0" else torch.bfloat16)
    if optimizers is None:
        return models
    return models, optimizers


@contextlib.contextmanager
def scale_loss(loss, optimizers, loss_id=0, model=None, delay_unscale=False, delay_overflow_check=False):
    yield loss          # loss scale 1 (bf16 range); gradients arrive unscaled in the fp32 master parameters


def master_params(optimizer):
    for g in optimizer.param_groups:
        for p in g["params"]:
            yield p


def state_dict():
    return dict(_state)


def load_state_dict(sd):
    _state.update({k: sd[k] for k in ("opt_level", "loss_scale") if k in sd})
    precision.set_low_precision(None if _state["opt_level"] == "O0" else torch.bfloat16)
