"""SVDHead: weighted Kabsch alignment (reference: rslo/layers/svd.py:7-64).

H = (src - mean(src)) diag(w) (tgt - mean(tgt))^T with UNWEIGHTED centroids, R = V U^T with the
reflection fixed on V's last column, and the function returns the INVERSE motion (R^T, -R^T t),
i.e. the map target -> source.  MI355X form: batched, branch-free (the reflection fix is the
diag(1,1,det) identity) so there is no per-sample Python loop and no host sync on det(R)."""
import apex.amp as amp
import torch
import torch.nn as nn


class SVDHead(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        reflect = torch.eye(3)
        reflect[2, 2] = -1
        self.reflect = nn.Parameter(reflect, requires_grad=False)

    @amp.float_function
    def forward(self, src, tgt, weight=None):
        """src, tgt: [B,3,N]; weight: [B,N] -> R [B,3,3], t [B,3]."""
        B = src.size(0)
        src_mean = src.mean(dim=2, keepdim=True)
        tgt_mean = tgt.mean(dim=2, keepdim=True)
        sc, tc = src - src_mean, tgt - tgt_mean
        if weight is not None:
            sc = sc * weight[:, None, :]
        H = torch.matmul(sc, tc.transpose(2, 1))
        R = kabsch_rotation(H)
        t = torch.matmul(-R, src_mean) + tgt_mean
        R = R.transpose(-1, -2).contiguous()
        t = -R @ t
        return R, t.view(B, 3)


def kabsch_rotation(H):
    """R = V diag(1, 1, det(V U^T)) U^T for H = U S V^T, batched [B,3,3]."""
    U, S, Vh = torch.linalg.svd(H)
    V = Vh.transpose(-1, -2)
    d = torch.det(V @ U.transpose(-1, -2))
    D = torch.diag_embed(torch.stack([torch.ones_like(d), torch.ones_like(d), torch.sign(d)], -1))
    return V @ D @ U.transpose(-1, -2)
