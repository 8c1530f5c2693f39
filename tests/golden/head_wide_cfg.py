"""Configuration of the wide-head fixture (tests/golden/head_wide.npz), shared by its generator and its tests."""
import numpy as np
import torch

CFG = dict(bn_type="SyncBN", enc_use_norm=True, conv_type="mask_conv", layer_nums=[1, 2, 1], layer_strides=[2, 2, 2],
           num_filters=[32, 32, 64], upsample_strides=[2, 2, 2], num_upsample_filters=[64, 64, 64], num_input_features=64,
           pooling_type="avg_pool", pooling_size=1, dropout=1e-22, cycle_constraint=True, pred_pyramid_motion=True,
           use_deep_supervision=True, odom_format="rx+t", dense_predict=True, conf_type="softmax", use_svd=False,
           cubic_pred_height=0)
PC_RANGE = np.array([-70.4, -38.4, -3, 70.4, 38.4, 5], np.float32)
SHAPE = (2, 32, 48, 64)
FULL_GRADS = ("blocks.0.0.conv1.conv1.weight", "blocks.0.0.downsample.0.conv1.weight", "blocks.1.1.conv2.conv1.weight",
              "blocks.2.0.conv1.conv1.weight", "skip_blocks.0.0.weight", "deblocks.2.1.weight", "tq_map_conv.0.weight",
              "tq_map_conv.6.weight", "t_map_conf.conf_model.3.weight", "blocks.1.0.bn1.weight", "blocks.1.0.bn1.bias")


def functional(res, seed=77):
    """A fixed linear functional of the head's differentiable outputs (coefficients from a seed)."""
    rs = np.random.RandomState(seed)
    total = 0.0
    outs = [res["translation_preds"][0], res["rotation_preds"][0], res["t_conf"], res["r_conf"]] + \
           [p[0] for p in res["pyramid_motion"]]
    for o in outs:
        c = torch.from_numpy(rs.standard_normal(tuple(o.shape)).astype(np.float32)).to(o.device)
        total = total + (o * c).sum() / float(np.sqrt(o.numel()))
    return total


