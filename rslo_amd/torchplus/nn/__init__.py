import torch


class Empty(torch.nn.Module):
    """Identity placeholder (reference: rslo/torchplus/nn/modules/common.py:8-18)."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *args, **kwargs):
        if len(args) == 1:
            return args[0]
        if len(args) == 0:
            return None
        return args


class Sequential(torch.nn.Sequential):
    pass


GroupNorm = torch.nn.GroupNorm
