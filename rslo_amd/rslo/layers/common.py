import torch
import torch.nn as nn


class ParameterLayer(nn.Module):
    """A bare learnable tensor exposed as `.param` (reference: rslo/layers/common.py)."""

    def __init__(self, init_value, requires_grad=True):
        super().__init__()
        self.param = nn.Parameter(torch.as_tensor(init_value).clone(), requires_grad=requires_grad)

    def forward(self):
        return self.param
