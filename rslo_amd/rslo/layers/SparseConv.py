"""(feature, mask)-pair aware normalisation / activation wrappers
(reference: rslo/layers/SparseConv.py:96-132,195-220): they act on the feature tensor and pass the
mask through untouched."""
import apex
import torch.nn as nn


def _pair(fn, x):
    if isinstance(x, (tuple, list)):
        return [fn(x[0]), x[1]]
    return fn(x)


class SPC_SyncBN2d(apex.parallel.SyncBatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None, channel_last=False, fuse_relu=False, noise_scale_std=0, noise_shift_std=0):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, process_group, channel_last)
        assert noise_scale_std == 0 and noise_shift_std == 0, "BN noise is not used by the RSLO hot path"
        self.add_noise = False

    def forward(self, x):
        return _pair(super().forward, x)


class SPC_BN2d(nn.BatchNorm2d):
    def forward(self, x):
        return _pair(super().forward, x)


class SPC_ReLU(nn.ReLU):
    def forward(self, x):
        return _pair(super().forward, x)


class SPC_LeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        return _pair(super().forward, x)
