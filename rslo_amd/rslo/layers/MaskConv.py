"""MaskConv: a bias-free Conv2d that carries an occupancy mask alongside the features
(reference: rslo/layers/MaskConv.py:20-73).  The features are NOT multiplied by the mask; the mask is
only propagated (max-pooled with the conv's geometry) for later use.  Dense conv2d goes to MIOpen."""
import torch
import torch.nn as nn

from rslo.layers.hip_conv2d import Conv2d


class MaskMaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if isinstance(x, (list, tuple)):
            return super().forward(x[0]), x[1]
        return super().forward(x)


class MaskConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True,
                 max_pool_mask=True, groups=1):
        super().__init__()
        assert max_pool_mask, "conv-propagated masks are not used by the RSLO hot path"
        self.out_channels = out_channels
        self.use_bias = bias
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, bias=False,
                               padding=padding, groups=groups)
        self.max_pool_mask = True
        self.mask_pool = nn.MaxPool2d(kernel_size, stride=stride, padding=padding)

    def forward(self, x):
        if not isinstance(x, (list, tuple)):
            x = [x, (x.abs().sum(dim=1, keepdim=True) != 0).float().detach()]
        tensor, mask = x
        if mask is not None:        # None: the caller declared the mask unused (odom_pred.py, `track_masks`)
            with torch.no_grad():
                mask = self.mask_pool(mask)
        return [self.conv1(tensor), mask]


class MaskConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        if not isinstance(x, (list, tuple)):
            x = [x, (x.sum(dim=1, keepdim=True) != 0).float().detach()]
        return super().forward(x[0]), x[1].detach()
