"""Dense BEV encoder-decoder shared by the odometry heads
(reference: rslo/models/odom_pred_base.py:24-324; architecture table SURVEY.md App-A.2).

Only the configuration the shipped prototxt selects is implemented (conv_type "mask_conv", BN family
normalisation, dense_predict); other registry options raise.  All dense conv2d / BN / pooling work is
plain torch -> MIOpen: it is MFMA-bound library work, not part of the hand-written kernel set.
Attribute and sub-module names equal the reference's so that state dicts are interchangeable.
"""
import torch
from torch import nn

from rslo.layers.common import ParameterLayer
from rslo.layers.confidence import ConfidenceModule
from rslo.layers.hip_conv2d import Conv2d
from rslo.layers.MaskConv import MaskConv, MaskConvTranspose2d
from rslo.layers.SparseConv import (SPC_BN2d, SPC_LeakyReLU, SPC_MaskSyncBN2d, SPC_ReLU, SPC_SemiGlobalSyncBN2d,
                                    SPC_SyncBN2d, FusedSequential)
from torchplus.nn import Empty
from torchplus.tools import change_default_args


def conf_trunk(cin, BatchNorm2d, ReLU):
    """64 -> 64 -> 32 -> 1 confidence trunk (odom_pred_base.py:250-276)."""
    return FusedSequential(Conv2d(cin, 64, kernel_size=3, padding=1), BatchNorm2d(64), ReLU(),
                         Conv2d(64, 32, kernel_size=3, padding=1), BatchNorm2d(32), ReLU(),
                         Conv2d(32, 1, kernel_size=1))


class OdomPredEncDecBase(nn.Module):
    def __init__(self, point_cloud_range=None, enc_use_norm=True, seq_len=1, layer_nums=(3, 5, 5),
                 layer_strides=(2, 2, 2), num_filters=(128, 128, 256), upsample_strides=(1, 2, 4),
                 num_upsample_filters=(256, 256, 256), num_input_features=128, encode_background_as_zeros=True,
                 use_groupnorm=False, bn_type="BN", num_groups=32, dropout=0.2, pooling_type="avg_pool",
                 pooling_size=1, cycle_constraint=False, conv_type="official", odom_format="rx+t",
                 pred_pyramid_motion=False, use_deep_supervision=False, use_loss_mask=True,
                 use_dynamic_mask=False, dense_predict=False, use_correlation=False, conf_type="linear",
                 use_SPGN=False, sync_bn=False, use_leakyReLU=False, dropout_input=False, first_conv_groups=1,
                 use_se=False, use_sa=False, use_svd=False, cubic_pred_height=0, name="odomPred", **kwargs):
        super().__init__()
        assert conv_type in ["official", "sparse_conv", "mask_conv"]
        assert odom_format in ["rx+t", "r(x+t)"]
        assert pooling_type in ["avg_pool", "max_pool"]
        assert bn_type in ["None", "BN", "IN", "SyncBN", "SemiGlobalSyncBN", "MaskSyncBN"]
        assert conf_type in ["linear", "softmax"]
        if conv_type != "mask_conv":
            raise NotImplementedError("only conv_type='mask_conv' is on the RSLO hot path")
        if bn_type == "IN" or use_groupnorm:
            raise NotImplementedError("bn_type %r is outside the RSLO hot path" % bn_type)
        if use_dynamic_mask or use_correlation or dropout_input or use_SPGN or use_se or use_sa:
            raise NotImplementedError("option outside the shipped RSLO configuration")
        layer_nums, layer_strides = list(layer_nums), list(layer_strides)
        num_filters, upsample_strides = list(num_filters), list(upsample_strides)
        num_upsample_filters = list(num_upsample_filters)

        self.conf_type = conf_type
        self._cubic_pred_height = cubic_pred_height
        self.point_cloud_range = point_cloud_range
        self.odom_format = odom_format
        self._use_sparse_conv, self._use_mask_conv = False, True
        self.dense_predict = dense_predict
        self._use_loss_mask = use_loss_mask
        self._use_dynamic_mask = False
        self._dropout_input = False
        self._first_conv_groups = first_conv_groups
        self.use_se = self.use_sa = False
        self.use_svd = use_svd
        self._bn_type = bn_type
        self._use_correlation = False
        self._enc_use_norm = enc_use_norm
        self._cycle_constraint = cycle_constraint
        self._num_input_features = num_input_features
        self._upsample_start_idx = len(layer_nums) - len(upsample_strides)
        self.pred_pyramid_motion = use_deep_supervision

        self.ReLU = SPC_ReLU if not use_leakyReLU else change_default_args(negative_slope=1e-3)(SPC_LeakyReLU)
        if bn_type == "None":
            self.BatchNorm2d = Empty
        else:
            # SyncBN is the shipped choice (fused kernels); the two statistics variants of the registry
            # (odom_pred_base.py:113-131) run on plain torch ops
            base = {"SemiGlobalSyncBN": SPC_SemiGlobalSyncBN2d, "MaskSyncBN": SPC_MaskSyncBN2d, "BN": SPC_BN2d}.get(
                bn_type, SPC_SyncBN2d)
            if sync_bn:
                base = SPC_SyncBN2d
            self.BatchNorm2d = change_default_args(eps=1e-3, momentum=0.01)(base)
        self.ConvTranspose2d = change_default_args(bias=True)(MaskConvTranspose2d)

        in_filters = [num_input_features, *num_filters[:-1]]
        blocks, skip_blocks, deblocks = [], [], []
        for i, layer_num in enumerate(layer_nums):
            block, num_out = self._make_layer(in_filters[i], num_filters[i], layer_num, stride=layer_strides[i],
                                              first_groups=first_conv_groups if i == 0 else 1,
                                              use_norm=self._enc_use_norm)
            blocks.append(block)
            if i - self._upsample_start_idx >= 0:
                skip_blocks.append(FusedSequential(Conv2d(num_out, num_out, kernel_size=3, stride=1, padding=1),
                                                 self.BatchNorm2d(num_out), self.ReLU()))
        for i, nuf in enumerate(num_upsample_filters):
            cin = num_filters[-1] * 2 if i == 0 else num_upsample_filters[i - 1] + num_filters[-(i + 1)]
            deblocks.append(FusedSequential(nn.Upsample(scale_factor=upsample_strides[i]),
                                          Conv2d(cin, nuf, kernel_size=3, stride=1, padding=1),
                                          self.BatchNorm2d(nuf), self.ReLU()))
        if self.pred_pyramid_motion:
            self.mask_gen_pools = nn.ModuleList(
                [nn.MaxPool2d(kernel_size=3, stride=s, padding=1) for s in upsample_strides])
        self._num_out_filters = cin
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        self.skip_blocks = nn.ModuleList(skip_blocks)
        # registration slot of the pyramid heads: subclasses fill it, its position fixes the parameter ORDER
        # (optimizer checkpoints index parameters by position; reference odom_pred.py registers it here)
        self.pyramid_motion_blocks = nn.ModuleList()

        last = num_upsample_filters[-1]
        self.tq_map_conv = FusedSequential(Conv2d(last, 64, kernel_size=3, padding=1), self.BatchNorm2d(64),
                                         self.ReLU(), Conv2d(64, 32, kernel_size=3, padding=1),
                                         self.BatchNorm2d(32), self.ReLU(), Conv2d(32, 7, kernel_size=1))
        self.q_map_conf = ConfidenceModule(conf_trunk(last, self.BatchNorm2d, self.ReLU), conf_type=conf_type)
        self.t_map_conf = ConfidenceModule(conf_trunk(last, self.BatchNorm2d, self.ReLU), conf_type=conf_type)

        self.pool = (nn.AdaptiveAvgPool2d if pooling_type == "avg_pool" else nn.AdaptiveMaxPool2d)(
            (pooling_size, pooling_size))
        self.fc1 = nn.Linear(num_filters[-1] * pooling_size * pooling_size * seq_len, 1024)
        assert dropout > 0
        self.odom_dropout = nn.Dropout(p=dropout)
        self.dense_dropout = nn.Dropout2d(p=dropout)
        self.fc2 = nn.Linear(1024, 7)
        self.softmax = nn.Softmax(dim=-1)
        self._use_SPGN = False
        self.SPGN = Empty()
        self.dynamic_sigma = ParameterLayer(torch.ones(1) * 0.1, requires_grad=True)

    def create_cycle_constraint_data(self, xs):
        """All (i < j) frame pairs: [x_i ...], [x_j ...], each [B * npairs, C, H, W]
        (pairs vary fastest within a batch element)."""
        assert len(xs) >= 2
        _, Cc, H, W = xs[0].shape
        first, second = [], []
        for i in range(len(xs)):
            for j in range(i + 1, len(xs)):
                first.append(xs[i])
                second.append(xs[j])
        if len(first) == 1:      # two frames: the stack + reshape of one element is the tensor itself
            return [first[0], second[0]]
        return [torch.stack(first, dim=1).reshape(-1, Cc, H, W), torch.stack(second, dim=1).reshape(-1, Cc, H, W)]

    def unravel_prediction(self, pred, seq_len):
        bs = pred.shape[0] // seq_len
        return torch.cat([pred[i * bs:(i + 1) * bs] for i in range(seq_len)], dim=1)
