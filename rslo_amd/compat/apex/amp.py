import contextlib
import functools

import torch


def float_function(fn):
    """apex marks fp32-only functions with this decorator; here: run the function with autocast off
    and floating inputs promoted to fp32 (what apex O1 does for registered float functions)."""

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if not torch.is_autocast_enabled():
            return fn(*args, **kwargs)

        def cast(x):
            if isinstance(x, torch.Tensor) and x.is_floating_point() and x.dtype != torch.float32:
                return x.float()
            if isinstance(x, (list, tuple)):
                return type(x)(cast(v) for v in x)
            return x

        with torch.autocast(device_type="cuda", enabled=False):
            return fn(*[cast(a) for a in args], **{k: cast(v) for k, v in kwargs.items()})

    return wrapper


half_function = promote_function = lambda fn: fn


def register_float_function(module, name):
    setattr(module, name, float_function(getattr(module, name)))


register_half_function = register_promote_function = lambda module, name: None


def initialize(models, optimizers=None, opt_level="O0", **kwargs):
    """O0 (the reference default, train_hdf5.py:456) is plain fp32: nothing to patch."""
    if optimizers is None:
        return models
    return models, optimizers


@contextlib.contextmanager
def scale_loss(loss, optimizers, **kwargs):
    yield loss


def master_params(optimizer):
    for g in optimizer.param_groups:
        for p in g["params"]:
            yield p
