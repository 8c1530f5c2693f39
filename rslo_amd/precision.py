"""Compute precision of the convolution operands (BASELINE config C4).

None (default)      fp32 everywhere: split-bf16 kernels, fp32-level accuracy.
torch.bfloat16      what `apex.amp.initialize(..., opt_level="O1")` turns on (rslo_amd/compat/apex/amp.py): the
                    convolutions -- the 32/64-channel sparse layers of the encoder trunk and the dense 3x3 layers of the BEV
                    head -- take bf16 operands with fp32 accumulation; master weights, BatchNorm, the vote, the
                    covariance branch, chamfer / ICP / loss maths stay fp32 (the reference's `@amp.float_function`
                    islands).  MI355X has no reason to use fp16: bf16 has the fp32 exponent range, so
                    no loss scaling is needed.
"""
import contextlib

import torch

_low = None


def low_precision():
    return _low


def set_low_precision(dtype):
    global _low
    if dtype not in (None, torch.bfloat16):
        raise ValueError("low precision must be None or torch.bfloat16 (fp16 is not a mode of this path)")
    _low = dtype


@contextlib.contextmanager
def low_precision_as(dtype):
    prev = _low
    set_low_precision(dtype)
    try:
        yield
    finally:
        set_low_precision(prev)
