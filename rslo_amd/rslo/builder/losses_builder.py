"""losses_builder.build (reference: rslo/builder/losses_builder.py:23-151) for the configured loss types."""
from rslo.core import losses


def _pose_loss(cfg):
    if cfg.loss_type == "AdaptiveWeightedL2":
        scale = cfg.balance_scale if cfg.balance_scale > 0 else 1
        return losses.AdaptiveWeightedL2Loss(cfg.init_alpha, learn_alpha=not cfg.not_learn_alpha,
                                             loss_weight=cfg.weight, focal_gamma=cfg.focal_gamma,
                                             balance_scale=scale)
    if cfg.loss_type in ("L2", "AdaptiveWeightedL2RMatrixLoss"):
        raise NotImplementedError("loss_type %r is not configured on the RSLO hot path" % cfg.loss_type)
    raise ValueError("Empty loss config.")


def _consistency_loss(cfg):
    if cfg.loss_type == "Aleat5_1ChamferL2NormalWeightedALLSVDLoss":
        assert cfg.penalize_ratio > 0 and cfg.pred_downsample_ratio > 0 and cfg.reg_weight > 0 and cfg.sph_weight > 0
        return losses.Aleat5_1ChamferL2NormalWeightedALLSVDLoss(
            loss_weight=cfg.weight, penalize_ratio=cfg.penalize_ratio, sample_block_size=cfg.sample_block_size,
            norm=cfg.norm, pred_downsample_ratio=cfg.pred_downsample_ratio, reg_weight=cfg.reg_weight,
            sph_weight=cfg.sph_weight)
    if cfg.loss_type == "AdaptiveWeightedL2":
        return losses.AdaptiveWeightedL2Loss(cfg.init_alpha, learn_alpha=not cfg.not_learn_alpha,
                                             loss_weight=cfg.weight, focal_gamma=cfg.focal_gamma)
    print("Warning: Empty loss config.")
    return None


def build(loss_config):
    """-> (rotation, translation, pyramid_rotation, pyramid_translation, consistency).  Unconfigured
    pyramid losses alias the global ones (same objects: their alpha is shared, losses_builder.py:40-50)."""
    rot = _pose_loss(loss_config.rotation_loss)
    trans = _pose_loss(loss_config.translation_loss)
    py_rot = _pose_loss(loss_config.pyramid_rotation_loss) if loss_config.pyramid_rotation_loss.loss_type != "" else rot
    py_trans = _pose_loss(loss_config.pyramid_translation_loss) \
        if loss_config.pyramid_translation_loss.loss_type != "" else trans
    cons = _consistency_loss(loss_config.consistency_loss)
    if loss_config.rigid_transform_loss.weight != 0:
        raise NotImplementedError("RigidTransformLoss is not configured on the RSLO hot path")
    return rot, trans, py_rot, py_trans, cons
