"""Generates tests/golden/chamfer_ref.npz from the REFERENCE's own chamfer CPU path
(/root/reference/thirdparty/chamfer_distance/chamfer_distance.cpp compiled by oracle/build_ref.py).
Run in the authoring container only:  python tests/golden/make_golden_chamfer.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

build_ref.build()
ref = build_ref.load()
rng = np.random.default_rng(20260929)
b, n, m = 2, 1500, 1300
a = (rng.normal(size=(b, n, 3)) * np.array([20, 10, 1.5])).astype(np.float32)
c = (rng.normal(size=(b, m, 3)) * np.array([20, 10, 1.5])).astype(np.float32)
c[:, :40] = a[:, 100:140]           # exact hits
c[:, 40:80] = c[:, :40]             # duplicated targets -> ties, lowest index must win
d1, d2 = torch.zeros(b, n), torch.zeros(b, m)
i1, i2 = torch.zeros(b, n, dtype=torch.int32), torch.zeros(b, m, dtype=torch.int32)
ref.forward(torch.from_numpy(a), torch.from_numpy(c), d1, d2, i1, i2)
gd = rng.normal(size=(b, n)).astype(np.float32)
g1, g2 = torch.zeros(b, n, 3), torch.zeros(b, m, 3)
ref.backward(torch.from_numpy(a), torch.from_numpy(c), g1, g2, torch.from_numpy(gd), torch.zeros(b, m), i1, i2)
# both directions (cd.forward / cd.backward as ChamferDistance / ChamferDistanceWithIdx use them,
# chamfer_distance.py:47-130): second-direction outputs and the gradient with BOTH upstream gradients non-zero
gd2 = rng.normal(size=(b, m)).astype(np.float32)
h1, h2 = torch.zeros(b, n, 3), torch.zeros(b, m, 3)
ref.backward(torch.from_numpy(a), torch.from_numpy(c), h1, h2, torch.from_numpy(gd), torch.from_numpy(gd2), i1, i2)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chamfer_ref.npz")
np.savez_compressed(out, xyz1=a, xyz2=c, dist1=d1.numpy(), idx1=i1.numpy(), graddist1=gd,
                    gradxyz1=g1.numpy(), gradxyz2=g2.numpy(), dist2=d2.numpy(), idx2=i2.numpy(), graddist2=gd2,
                    gradxyz1_both=h1.numpy(), gradxyz2_both=h2.numpy())
print("wrote", out, os.path.getsize(out))
