"""Golden vectors of the BEV head at widths the hand-written dense kernels take (channel counts that are multiples of
32), produced by RUNNING THE REFERENCE'S OWN head class from /root/reference (authoring container only):

    python tests/golden/make_golden_head_wide.py        ->  tests/golden/head_wide.npz

`head_small.npz` (8 / 16 channels) pins the head's graph but none of its convolutions can reach csrc/conv2d.hip
(32-channel granularity), so its GPU test runs them on the library.  Here `UNRResNetOdomPredEncDecSVDTempMask`
(rslo/models/odom_pred.py:360-426) is built with num_filters [32, 32, 64], 64 upsample filters and 2 x 32 input
channels on 48 x 64 maps: every 3x3 layer (stride 1 and 2), the 1x1 downsamples, the SyncBN layers, the vote and the
pyramid heads then run on the HIP kernels in the mirror, forward AND backward, against numbers the reference produced.
Stored: outputs, BatchNorm running statistics after the step, and gradients of a seeded linear functional of the
outputs (full tensors for a few layers, sum / abs-sum for all).  Weights and inputs are not stored: both sides rebuild
them from seeds (tests/golden/golden_weights.py).  Shims: those of make_golden_ref.py (apex / kornia stand-ins)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref as M          # noqa: E402  (install_shims puts /root/reference on the path)
from golden_weights import fill_dense_parameters, seeded_inputs   # noqa: E402

from head_wide_cfg import CFG, FULL_GRADS, PC_RANGE, SHAPE, functional   # noqa: E402  (shared with tests/test_head_wide.py)


def main():
    M.install_shims()
    import rslo.models.odom_pred as OP
    head = OP.get_odom_class("UNRResNetOdomPredEncDecSVDTempMask")(point_cloud_range=PC_RANGE, **CFG)
    fill_dense_parameters(head, 5)
    head.train()
    xs = seeded_inputs(9, 3, SHAPE)
    res = head([x.clone() for x in xs])
    functional(res).backward()
    np_ = M.np_
    out = {"t_pred": np_(res["translation_preds"][0]), "r_pred": np_(res["rotation_preds"][0]),
           "tq_map_g": np_(res["tq_map_g"]), "t_conf": np_(res["t_conf"]), "r_conf": np_(res["r_conf"])}
    for i, p in enumerate(res["pyramid_motion"]):
        out["py%d_pred" % i], out["py%d_mask" % i] = np_(p[0]), np_(p[1])
    for k, v in head.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["sd_after/" + k] = np_(v)
    names, sums = [], []
    for n, p in head.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        sums.append([float(p.grad.double().sum()), float(p.grad.double().abs().sum()), float(p.grad.abs().max())])
        if n in FULL_GRADS:
            out["grad/" + n] = np_(p.grad)
    out["grad_names"] = np.array(names)
    out["grad_sums"] = np.array(sums, np.float64)
    np.savez_compressed(os.path.join(HERE, "head_wide.npz"), **out)
    print("wrote head_wide.npz:", len(names), "gradient tensors,", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) // 1024, "KB raw")


if __name__ == "__main__":
    main()
