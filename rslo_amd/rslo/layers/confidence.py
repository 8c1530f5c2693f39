"""ConfidenceModule: per-cell vote weights (reference: rslo/layers/confidence.py:5-37).
softmax variant: masked cells get logit -1000 (not -inf), softmax over the flattened H*W per channel."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConfidenceModule(nn.Module):
    def __init__(self, conf_model, conf_type="softmax"):
        super().__init__()
        assert conf_type in ["linear", "softmax"]
        self.conf_model = conf_model
        self.conf_type = conf_type
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x, extra_mask=None, temperature=1, return_logit=False):
        logit = self.conf_model(x)
        if extra_mask is None:
            extra_mask = torch.ones_like(logit)
        if self.conf_type == "linear":
            conf = (F.elu(logit) + 1 + 1e-12) * (extra_mask + 1e-12)
        else:
            masked = torch.where(extra_mask > 0, logit, torch.full_like(logit, -1000))
            shp = masked.shape
            conf = F.softmax(masked.reshape(shp[0], shp[1], -1) / temperature, dim=-1).reshape(shp)
        if return_logit:
            return conf, logit
        return conf
