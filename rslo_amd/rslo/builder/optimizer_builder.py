"""Optimizer from the `train_config.optimizer` message (reference: rslo/builder/optimizer_builder.py:25-128).

Layer groups (optimizer_builder.py:49-66): [VFE], [middle], [odom head], [losses], each flattened to its leaf
modules and split into (non-BN, BN) param groups by OptimWrapper -> 8 Adam groups.  `fixed_weight_decay` selects
decoupled decay with betas (0.9, 0.99).  On GPU parameters the inner torch optimizer runs fused (one multi-tensor
launch per group); RSLO_FUSED_OPTIM=0 turns that off.
"""
import os
from functools import partial

import torch
from torch import nn

from torchplus.train.fastai_optim import OptimWrapper


def flatten_model(m):
    """Leaf modules of `m` in registration order."""
    if m is None:
        return []
    kids = list(m.children())
    if not kids:
        return [m]
    out = []
    for c in kids:
        out += flatten_model(c)
    return out


def get_layer_groups(m):
    return [nn.ModuleList(flatten_model(m))]


def get_voxeLO_net_layer_groups(net):
    losses = nn.Sequential(net._rotation_loss, net._translation_loss, net._pyramid_rotation_loss,
                           net._pyramid_translation_loss, net._consistency_loss)
    return [get_layer_groups(net.voxel_feature_extractor), get_layer_groups(net.middle_feature_extractor),
            get_layer_groups(net.odom_predictor), get_layer_groups(losses)]


def _which(msg, oneof):
    if hasattr(msg, "WhichOneof"):
        return msg.WhichOneof(oneof)
    raise ValueError("optimizer config must support WhichOneof")


def build(optimizer_config, net, name=None, mixed=False, loss_scale=512.0):
    if mixed:
        raise NotImplementedError("mixed=True (fp16 master weights) is not used by the RSLO path")
    kind = _which(optimizer_config, "optimizer")
    on_gpu = any(p.is_cuda for p in net.parameters())
    fused = {"fused": True} if on_gpu and os.environ.get("RSLO_FUSED_OPTIM", "1") != "0" else {}
    if kind == "rms_prop_optimizer":
        cfg = optimizer_config.rms_prop_optimizer
        opt_func = partial(torch.optim.RMSprop, alpha=cfg.decay, momentum=cfg.momentum_optimizer_value,
                           eps=cfg.epsilon)
    elif kind == "momentum_optimizer":
        cfg = optimizer_config.momentum_optimizer
        opt_func = partial(torch.optim.SGD, momentum=cfg.momentum_optimizer_value)
    elif kind == "adam_optimizer":
        cfg = optimizer_config.adam_optimizer
        if optimizer_config.fixed_weight_decay:
            opt_func = partial(torch.optim.Adam, betas=(0.9, 0.99), amsgrad=cfg.amsgrad, **fused)
        else:
            opt_func = partial(torch.optim.Adam, amsgrad=cfg.amsgrad, **fused)
    else:
        raise ValueError("Optimizer %s not supported." % kind)
    optimizer = OptimWrapper.create(opt_func, 3e-3, get_voxeLO_net_layer_groups(net), wd=cfg.weight_decay,
                                    true_wd=optimizer_config.fixed_weight_decay, bn_wd=True)
    if optimizer_config.use_moving_average:
        raise ValueError("torch don't support moving average")
    optimizer.name = kind if name is None else name
    return optimizer
