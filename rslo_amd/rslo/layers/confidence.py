"""ConfidenceModule: per-cell vote weights of the ego-motion vote (reference: rslo/layers/confidence.py:5-37).

"softmax" type: cells outside `extra_mask` get the logit -1000 (a large finite value, not -inf), then a softmax
with temperature runs over the flattened H*W cells of every (sample, channel).  "linear" type: (elu + 1) scaled by
the mask.  The trunk (`conf_model`) is an ordinary dense conv stack.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

MASKED_LOGIT = -1000.0


def masked_spatial_softmax(logit, mask, temperature=1.0, outside=None):
    """softmax over dims (H, W) of [B, C, H, W]; positions with mask <= 0 take MASKED_LOGIT first.
    outside: optional precomputed boolean `~(mask > 0)` (the head evaluates four softmaxes over one mask)."""
    b, c = logit.shape[:2]
    if outside is None:
        outside = ~(mask > 0)
    z = logit.masked_fill(outside.expand_as(logit), MASKED_LOGIT).reshape(b, c, -1)
    if temperature != 1:
        z = z / temperature
    return F.softmax(z, dim=-1).reshape(logit.shape)


class ConfidenceModule(nn.Module):
    def __init__(self, conf_model, conf_type="softmax"):
        super().__init__()
        if conf_type not in ("linear", "softmax"):
            raise AssertionError("conf_type must be 'linear' or 'softmax'")
        self.conf_model = conf_model
        self.conf_type = conf_type
        self.softmax = nn.Softmax(dim=-1)      # kept for attribute parity; parameter-free

    def forward(self, x, extra_mask=None, temperature=1, return_logit=False, outside=None):
        logit = self.conf_model(x)
        mask = torch.ones_like(logit) if extra_mask is None else extra_mask
        if self.conf_type == "softmax":
            conf = masked_spatial_softmax(logit, mask, temperature, outside)
        else:
            conf = (F.elu(logit) + 1 + 1e-12) * (mask + 1e-12)
        return (conf, logit) if return_logit else conf
