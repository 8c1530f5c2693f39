import torch


def roll(x, shift, dim=-1):
    """Cyclic shift along `dim` (reference: rslo/torchplus/ops/array_ops.py:34-53); used to move
    between RSLO's (w,x,y,z) and kornia's (x,y,z,w) quaternion layouts."""
    return torch.roll(x, shifts=int(shift), dims=dim)
