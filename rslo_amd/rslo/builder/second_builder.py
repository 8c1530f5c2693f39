"""second_builder.build (reference: rslo/builder/second_builder.py:26-137): config message -> network.
Passes the same ~60 kwargs to the registered network class."""
from rslo.builder import losses_builder
from rslo.models.voxel_odom_net import get_voxelnet_class


def build(model_cfg, voxel_generator, measure_time=False, testing=False):
    vfe, mid, od = model_cfg.voxel_feature_extractor, model_cfg.middle_feature_extractor, model_cfg.odom_predictor
    vfe_num_filters = list(vfe.num_filters)
    grid_size = voxel_generator.grid_size
    dense_shape = [1] + grid_size[::-1].tolist() + [vfe_num_filters[-1]]
    rot, trans, py_rot, py_trans, cons = losses_builder.build(model_cfg.loss)
    return get_voxelnet_class(model_cfg.network_class_name)(
        dense_shape, pc_range=voxel_generator.point_cloud_range,
        vfe_class_name=vfe.module_class_name, vfe_num_filters=vfe_num_filters,
        middle_class_name=mid.module_class_name, middle_num_input_features=mid.num_input_features,
        middle_num_filters_d1=list(mid.num_filters_down1), middle_num_filters_d2=list(mid.num_filters_down2),
        middle_use_leakyReLU=mid.use_leakyReLU, middle_relu_type=mid.relu_type,
        odom_class_name=od.module_class_name, odom_num_input_features=od.num_input_features,
        odom_layer_nums=list(od.layer_nums), odom_layer_strides=list(od.layer_strides),
        odom_num_filters=list(od.num_filters), odom_upsample_strides=list(od.upsample_strides),
        odom_num_upsample_filters=list(od.num_upsample_filters), odom_pooling_size=od.pool_size,
        odom_pooling_type=od.pool_type, odom_cycle_constraint=od.cycle_constraint, odom_conv_type=od.conv_type,
        odom_format=od.odom_format, odom_pred_pyramid_motion=od.pred_pyramid_motion,
        odom_use_deep_supervision=od.use_deep_supervision, odom_use_loss_mask=not od.not_use_loss_mask,
        odom_use_dynamic_mask=od.use_dynamic_mask, odom_dense_predict=od.dense_predict, odom_use_corr=od.use_corr,
        odom_dropout=od.dropout, odom_conf_type=od.conf_type, odom_use_SPGN=od.use_SPGN,
        odom_use_leakyReLU=od.use_leakyReLU, vfe_use_norm=not vfe.not_use_norm, middle_bn_type=mid.bn_type,
        odom_bn_type=od.bn_type, odom_enc_use_norm=not od.not_use_enc_norm, odom_dropout_input=od.dropout_input,
        odom_first_conv_groups=max(1, od.first_conv_groups), odom_use_se=od.odom_use_se,
        odom_use_sa=od.odom_use_sa, odom_use_svd=od.use_svd, odom_cubic_pred_height=od.cubic_pred_height,
        freeze_bn=model_cfg.freeze_bn, freeze_bn_affine=model_cfg.freeze_bn_affine,
        freeze_bn_start_step=model_cfg.freeze_bn_start_step, use_GN=model_cfg.use_GN,
        num_input_features=model_cfg.num_point_features,
        encode_background_as_zeros=model_cfg.encode_background_as_zeros, with_distance=vfe.with_distance,
        rotation_loss=rot, translation_loss=trans, pyramid_rotation_loss=py_rot,
        pyramid_translation_loss=py_trans, rigid_transform_loss=None, pyramid_rigid_transform_loss=None,
        consistency_loss=cons, measure_time=measure_time, voxel_generator=voxel_generator,
        pyloss_exp_w_base=model_cfg.loss.pyloss_exp_w_base, testing=testing, icp_iter=model_cfg.icp_iter)
