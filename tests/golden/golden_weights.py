"""Deterministic parameter values shared by the golden generators and the tests that replay them.

Fixtures stay small because weights are not stored: both sides rebuild them from a seed with
numpy's legacy `RandomState` (bit-stable across numpy versions by policy).  Parameters are visited in
`named_parameters()` order -- pinned identical between the reference and the mirror by
tests/golden/state_dict_keys.json.
"""
import numpy as np
import torch


def fill_parameters(module, seed, gain=1.0):
    """Sparse-conv weights [kz,ky,kx,Cin,Cout]: N(0, gain^2 * 2/(0.4*K*Cin)) (about unit activations on a
    KITTI-like occupancy of ~40 % of the 27 taps); biases N(0, 0.1^2); BatchNorm gamma 1 + N(0, 0.1^2)."""
    with torch.no_grad():
        for i, (name, p) in enumerate(module.named_parameters()):
            rs = np.random.RandomState(seed * 1000 + i)
            v = rs.standard_normal(tuple(p.shape)).astype(np.float32)
            if p.dim() == 5:
                k = p.shape[0] * p.shape[1] * p.shape[2]
                v *= np.float32(gain * np.sqrt(2.0 / (0.4 * k * p.shape[3])))
            elif name.endswith("weight"):       # BatchNorm gamma
                v = np.float32(1.0) + np.float32(0.1) * v
            else:
                v *= np.float32(0.1)
            p.copy_(torch.from_numpy(v))


def fill_dense_parameters(module, seed):
    """Dense head: Conv2d weights [Cout,Cin,k,k] ~ N(0, 2/(k k Cin)) (unit-variance activations through ReLU), 1-D
    tensors named *.weight (BatchNorm gamma) 1 + N(0, 0.1^2), every other tensor (biases, scalars) N(0, 0.1^2)."""
    with torch.no_grad():
        for i, (name, p) in enumerate(module.named_parameters()):
            rs = np.random.RandomState(seed * 1000 + i)
            v = rs.standard_normal(tuple(p.shape)).astype(np.float32)
            if p.dim() == 4:
                v *= np.float32(np.sqrt(2.0 / (p.shape[1] * p.shape[2] * p.shape[3])))
            elif p.dim() == 1 and name.endswith("weight"):
                v = np.float32(1.0) + np.float32(0.1) * v
            else:
                v *= np.float32(0.1)
            p.copy_(torch.from_numpy(v))


def seeded_inputs(seed, n, shape, keep=0.4):
    """n sparse BEV-like maps [B,C,H,W]: N(0,1) values on a random `keep` fraction of the cells, zeros elsewhere."""
    out = []
    for i in range(n):
        rs = np.random.RandomState(seed * 100 + i)
        x = rs.standard_normal(shape).astype(np.float32)
        m = (rs.random_sample((shape[0], 1, shape[2], shape[3])) < keep).astype(np.float32)
        out.append(torch.from_numpy(x * m))
    return out
