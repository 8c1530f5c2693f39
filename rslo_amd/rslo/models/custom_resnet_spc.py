"""BasicBlock of the BEV encoder (reference: rslo/models/custom_resnet_spc.py:224-298): two 3x3
(mask-)convs with BN, residual add where masks are averaged, ReLU."""
import torch
import torch.nn as nn

from rslo.layers.SparseConv import SPC_LeakyReLU, SPC_ReLU, SPC_SyncBN2d, act_slope_of


def conv1x1(in_planes, out_planes, stride=1, Conv2d=None, groups=1):
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=0, bias=False, groups=groups)


def conv3x3(in_planes, out_planes, stride=1, Conv2d=None, groups=1):
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False, groups=groups)


def _mask_mean(ma, mb):
    if ma is None or mb is None:        # mask declared unused by the caller (MaskConv passes None through)
        return None
    return ((ma + mb) / 2).float()


def SPC_add(a, b):
    if isinstance(a, (list, tuple)):
        return [a[0] + b[0], _mask_mean(a[1], b[1])]
    return a + b


def SPC_cat(a, b):
    if isinstance(a, (list, tuple)):
        return [torch.cat([a[0], b[0]], dim=1), _mask_mean(a[1], b[1])]
    return torch.cat([a, b], dim=1)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, lrelu=False, BN=None, Conv2d=None, groups=1,
                 use_se=False, use_sa=False):
        super().__init__()
        assert not use_se and not use_sa, "SE / spatial attention are off in the shipped config"
        self.BatchNorm2d = BN if BN is not None else nn.BatchNorm2d
        self.Conv2d = Conv2d if Conv2d is not None else nn.Conv2d
        self.conv1 = conv3x3(inplanes, planes, stride, Conv2d=self.Conv2d, groups=groups)
        self.bn1 = self.BatchNorm2d(planes)
        self.relu = SPC_LeakyReLU(0.1, True) if lrelu else SPC_ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes, Conv2d=self.Conv2d, groups=groups)
        self.bn2 = self.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.use_se = self.use_sa = False

    def forward(self, x):
        if isinstance(self.bn1, SPC_SyncBN2d) and isinstance(self.bn2, SPC_SyncBN2d):
            # BN + ReLU and BN + residual add + ReLU as fused epilogues of the normalisation kernels
            slope = act_slope_of(self.relu)
            out = self.bn1(self.conv1(x), act_slope=slope)
            out = self.conv2(out)
            residual = x if self.downsample is None else self.downsample(x)
            y = self.bn2(out, act_slope=slope, residual=residual)
            if isinstance(y, (list, tuple)):      # (feature, mask) pairs: masks are averaged like SPC_add does
                y = [y[0], _mask_mean(out[1], residual[1])]
            return y
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(SPC_add(out, residual))
