"""ctypes binding of librslo_hip.so (include/rslo_hip.h) for torch tensors on a ROCm device.

This is the only place where Python meets the C ABI.  Tensors are passed as raw device
pointers plus sizes; work is enqueued on torch's current HIP stream.  There is NO fallback:
if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSLO_HIP_LIB") or os.path.join(_HERE, "librslo_hip.so")   # override: kernel experiments only
_lib = None

_I3 = C.c_int32 * 3
_F3 = C.c_float * 3
_F6 = C.c_float * 6

# name -> (restype, argtypes); checked against include/rslo_hip.h by tests/test_cabi.py
_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
SIGNATURES = {
    "rslo_abi_version": (C.c_int, []),
    "rslo_last_error": (C.c_char_p, []),
    "rslo_voxelize_ws_bytes": (_sz, [_i64]),
    "rslo_voxelize": (C.c_int, [_vp, _i64, _i, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "rslo_vfe_mean": (C.c_int, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "rslo_hash_capacity": (_i64, [_i64]),
    "rslo_hash_build": (C.c_int, [_vp, _i64, _i, _vp, _vp, _vp, _i64, _vp]),
    "rslo_rulebook_subm": (C.c_int, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "rslo_conv_bitmap_words": (_i64, [_i, _vp]),
    "rslo_scan_ws_bytes": (_sz, [_i64]),
    "rslo_conv_out_count": (C.c_int, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp, _vp]),
    "rslo_conv_out_coords": (C.c_int, [_vp, _vp, _i64, _i, _vp, _vp, _i64, _vp]),
    "rslo_rulebook_conv": (C.c_int, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "rslo_rulebook_conv_T": (C.c_int, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "rslo_spconv_fwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _f, _vp, _vp]),
    "rslo_spconv_dgrad": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "rslo_weight_transpose": (C.c_int, [_vp, _i, _i, _i, _vp, _vp]),
    "rslo_weight_split_bytes": (_sz, [_i, _i, _i]),
    "rslo_weight_split": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rslo_weight_split_many": (C.c_int, [_vp, _i, _i64, _vp]),
    "rslo_spconv_fwd_split": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, C.c_float, _vp, _vp]),
    "rslo_spconv_set_tiling": (None, [_i, _i]),
    "rslo_spconv_set_live_rows": (None, [_vp]),
    "rslo_peer_create_host": (C.c_int, [C.c_char_p, _i, _i, _i, C.POINTER(C.c_void_p)]),
    "rslo_peer_host_unlink": (C.c_int, [_vp]),
    "rslo_peer_ipc_handle_bytes": (C.c_int, []),
    "rslo_peer_create_device_begin": (C.c_int, [_i, _i, _i, C.POINTER(C.c_void_p), _vp]),
    "rslo_peer_create_device_finish": (C.c_int, [_vp, _vp]),
    "rslo_peer_set_timeout_ms": (C.c_int, [_vp, _i]),
    "rslo_peer_allreduce_f64": (C.c_int, [_vp, _vp, _i, _vp]),
    "rslo_peer_status": (C.c_ulonglong, [_vp, C.POINTER(C.c_int)]),
    "rslo_peer_destroy": (C.c_int, [_vp]),
    "rslo_tuning_set": (C.c_int, [C.c_char_p, _i]),
    "rslo_tuning_get": (C.c_int, [C.c_char_p, _vp]),
    "rslo_tuning_name": (C.c_char_p, [_i]),
    "rslo_weight_to_bf16": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rslo_spconv_fwd_bf16": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, C.c_float, _vp, _vp]),
    "rslo_spconv_wgrad_ws_bytes": (_sz, [_i64, _i, _i, _i]),
    "rslo_spconv_wgrad": (C.c_int, [_vp, _i, _vp, _i, _vp, _i64, _i, _vp, _sz, _vp, _vp, _vp]),
    "rslo_rulebook_row_order": (C.c_int, [_vp, _i64, _i, _i, _vp, _vp]),
    "rslo_leaky_bwd_colsum_bf16_blocks": (_i64, [_i64, _i]),
    "rslo_leaky_bwd_colsum_bf16": (C.c_int, [_vp, _vp, _i64, _i, C.c_float, _vp, _vp, _vp]),
    "rslo_spconv_wgrad_pairs_bf16": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i64, _i, _vp, _sz, _vp, _vp, _vp, _i,
                                               _vp]),
    "rslo_rulebook_pairs_ws_bytes": (_sz, [_i64, _i]),
    "rslo_rulebook_pairs": (C.c_int, [_vp, _i64, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "rslo_spconv_wgrad_pairs_ws_bytes": (_sz, [_i64, _i, _i, _i]),
    "rslo_spconv_wgrad_pairs": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i64, _i, _vp, _sz, _vp, _vp, _vp, _i, _vp]),
    "rslo_leaky_bwd_colsum_blocks": (_i64, [_i64, _i]),
    "rslo_leaky_bwd_colsum": (C.c_int, [_vp, _vp, _i64, _i, C.c_float, _vp, _vp, _vp]),
    "rslo_leaky_bwd": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp]),
    "rslo_segbn_ws_bytes": (_sz, [_i, _i64, _i]),
    "rslo_segbn_fwd": (C.c_int, [_vp, _i, _vp, _i, _i64, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _sz, _vp, _vp, _vp, _vp]),
    "rslo_segbn_bwd": (C.c_int, [_vp, _vp, _vp, _i, _vp, _i, _i64, _vp, _vp, _vp, _f, _vp, _sz, _vp, _vp, _vp, _vp]),
    "rslo_dense_scatter": (C.c_int, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "rslo_dense_gather": (C.c_int, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "rslo_dense_scatter_frames": (C.c_int, [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "rslo_dense_gather_frames": (C.c_int, [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "rslo_bev_channel_sums": (C.c_int, [_vp, _i, _i, _i, _i64, _vp, _vp]),
    "rslo_chamfer_ws_bytes": (_sz, [_i, _i, _i]),
    "rslo_chamfer_nn": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "rslo_chamfer_nn_ragged": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rslo_chamfer_brute_nn": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rslo_chamfer_grid_ws_bytes": (_sz, [_i, _i, _i]),
    "rslo_chamfer_grid_nn": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rslo_chamfer_grad": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rslo_cov_residual_ws_bytes": (_sz, [_i, _i]),
    "rslo_cov_residual_fwd": (C.c_int, [_vp] * 8 + [_i, _i, _i, _f, _vp, _sz, _vp, _vp, _vp]),
    "rslo_cov_residual_bwd": (C.c_int, [_vp] * 10 + [_i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rslo_cov_residual_bwd_ws_bytes": (_sz, [_i, _i, _i]),
    "rslo_icp_ws_bytes": (_sz, [_i, _i]),
    "rslo_icp_step": (C.c_int, [_vp] * 6 + [_i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "rslo_icp_step_first": (C.c_int, [_vp] * 6 + [_i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "rslo_transform_points": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "rslo_transform_rows": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "rslo_transform_rows_bwd_ws_bytes": (_sz, [_i, _i]),
    "rslo_transform_rows_bwd": (C.c_int, [_vp, _i, _vp, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "rslo_bn2d_ws_bytes": (_sz, [_i, _i, _i]),
    "rslo_bn2d_stats": (C.c_int, [_vp, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "rslo_bn2d_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp,
                                  _vp, _vp, _vp]),
    "rslo_bn2d_bwd_reduce": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp, _sz, _vp, _vp, _vp, _vp,
                                       _vp]),
    "rslo_bn2d_bwd_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _i, _i, _i, C.c_float, _i, _vp,
                                      _vp, _vp]),
    "rslo_bn2d_fwd_local": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _sz, _vp]),
    "rslo_bn2d_bwd_local": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp, _vp, _vp, _vp, _vp,
                                      _sz, _vp]),
    "rslo_roi_threshold": (C.c_int, [_vp, _i, _i, _vp, C.c_double, _vp, _vp]),
    "rslo_vote_ws_bytes": (_sz, [_i, _i, _i]),
    "rslo_vote_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "rslo_vote_bwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rslo_conv2d_wgrad_supported": (C.c_int, [_i, _i, _i, _i, _i]),
    "rslo_conv2d_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "rslo_conv1x1s2_wgrad_supported": (C.c_int, [_i, _i, _i, _i]),
    "rslo_conv1x1s2_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "rslo_conv1x1s2_wgrad": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "rslo_conv2d_wgrad": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "rslo_conv2d_fwd_supported": (C.c_int, [_i, _i, _i, _i]),
    "rslo_conv2d_wsplit_bytes": (_sz, [_i, _i]),
    "rslo_conv2d_wsplit": (C.c_int, [_vp, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_wsplit_k": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_s2_supported": (C.c_int, [_i, _i, _i]),
    "rslo_conv2d_fwd_s2": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_dgrad_s2": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_wsplit_many": (C.c_int, [_vp, _i, _i64, _vp]),
    "rslo_conv2d_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_fwd_bf16": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_cat_upsample_fwd": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_cat_upsample_bwd": (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rslo_bev_channel_sums_masks": (C.c_int, [_vp, _i, _i, _i, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "rslo_pose_tail_fwd": (C.c_int, [_vp, _i, _vp, _vp, _vp]),
    "rslo_pose_tail_bwd": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp]),
    "rslo_bev_display_ws_bytes": (_sz, [_i]),
    "rslo_bev_display": (C.c_int, [_vp, _i, _i, _i, C.c_int64, _vp, _vp, _vp, _sz, _vp]),
    "rslo_conv2d_dgrad_s2_add": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_fwd_add": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_fwd_add_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_wgrad_bf16": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "rslo_conv1x1_supported": (C.c_int, [_i, _i]),
    "rslo_conv1x1_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv1x1_dgrad": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv1x1_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "rslo_conv1x1_wgrad": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "rslo_quat_to_rot": (C.c_int, [_vp, _i, _vp, _vp]),
    "rslo_quat_to_rot_bwd": (C.c_int, [_vp, _vp, _i, _vp, _vp]),
    "rslo_pose_targets": (C.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "rslo_pose_targets_tq": (C.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "rslo_pair_rows_fwd": (C.c_int, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "rslo_pad_rows_fwd": (C.c_int, [_vp, _i64, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "rslo_pad_rows_bwd": (C.c_int, [_vp, _i64, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "rslo_pyramid_l2_ws_bytes": (_sz, [_vp, _i, _i]),
    "rslo_pyramid_l2_fwd": (C.c_int, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "rslo_pyramid_l2_bwd": (C.c_int, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rslo_plan_encoder_layout": (C.c_int, [_vp, _i, _vp, _vp]),
    "rslo_plan_encoder_pad_tails": (C.c_int, [_vp, _vp, _vp, _vp]),
    "rslo_plan_encoder": (C.c_int, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "rslo_tq_normalize_fwd": (C.c_int, [_vp, _i, _i64, _vp, _vp]),
    "rslo_tq_normalize_bwd": (C.c_int, [_vp, _vp, _i, _i64, _vp, _vp]),
    "rslo_conf_softmax_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "rslo_conf_softmax_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "rslo_head_masks_fwd": (C.c_int, [_vp, _vp]),
    "rslo_head_masks_bwd": (C.c_int, [_vp, _vp]),
    "rslo_loss_tail_fwd": (C.c_int, [_vp, _vp, _vp]),
    "rslo_loss_tail_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rslo_peer_wait_samples": (C.c_int, [_vp, _vp, _i]),
    "rslo_bn1d_eval_act": (C.c_int, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "rslo_wgrad_reduce_defer": (C.c_int, [_vp, _i, _vp]),
    "rslo_wgrad_reduce_many": (C.c_int, [_vp, _i, _vp]),
    "rslo_peer_capture_begin": (C.c_int, [_vp]),
    "rslo_peer_capture_end": (C.c_int, [_vp, _vp]),
    "rslo_peer_replay_prepare": (C.c_int, [_vp, _i, _vp]),
    "rslo_bn2d_peer_supported": (C.c_int, [_i, _i, _i]),
    "rslo_bn2d_fwd_peer": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp]),
    "rslo_bn2d_bwd_peer": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp, _vp, _vp, _vp,
                                     _vp]),
    "rslo_opl_bytes": (_sz, [_i, _i, _i, _i]),
    "rslo_opl_from_nchw": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rslo_conv2d_fwd_p_supported": (C.c_int, [_i, _i, _i, _i]),
    "rslo_conv2d_fwd_p": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rslo_opt_clip_grad_norm": (C.c_int, [_vp, _vp, _i, _f, _vp, _vp, _vp]),
    "rslo_opt_adam_step": (C.c_int, [_vp, _vp, _i, _vp, _f, _vp]),
}


class RsloHipError(RuntimeError):
    pass


def lib():
    """Load librslo_hip.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsloHipError(
                "librslo_hip.so not found at %s -- build it with `python -m rslo_amd.build` "
                "(there is no CPU fallback for the RSLO hot path)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib if _proxy is None else _proxy


_proxy = None


def probe_streams(on, points_only=None):
    """bench.py: while on, lib() hands out streamprobe.LibProxy -- every launching entry point is bracketed by HIP events on
    its stream (rslo_amd/streamprobe.py).  Off (the default, and during every timed step): the plain library."""
    global _proxy
    from rslo_amd import streamprobe
    if on:
        _proxy = streamprobe.LibProxy(lib() if _proxy is None else _proxy._lib, SIGNATURES, points_only)
        streamprobe.start()
    else:
        streamprobe.stop()
        _proxy = None


def tuning_set(name, value):
    """rslo_tuning_set: an explicit switch of the launch code (the library reads no environment variable)."""
    _chk(lib().rslo_tuning_set(name.encode(), int(value)), "rslo_tuning_set")


def tuning_get(name):
    v = C.c_int(0)
    _chk(lib().rslo_tuning_get(name.encode(), C.byref(v)), "rslo_tuning_get")
    return v.value


class tuning:
    """with capi.tuning(conv2d_fwd_kc=2, ...): switches set for the block and restored afterwards (tests, A/B scripts)."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = tuning_get(k)
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning_set(k, v)
        return False


def _chk(rc, name):
    if rc != 0:
        raise RsloHipError("%s failed (%d): %s" % (name, rc, lib().rslo_last_error().decode()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    # raw hipStream_t of torch's current stream: two C calls (torch.cuda.current_device() re-checks the lazy initialisation in
    # Python on every call: 1.5 us x ~270 launches per step); falls back to the public API
    # (a plain int is what a c_void_p parameter wants: no wrapper object per launch)
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t, dtype=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RsloHipError("%s must live on the GPU (the RSLO hot path has no CPU fallback)" % name)
    if dtype is not None and t.dtype != dtype:
        raise RsloHipError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RsloHipError("%s must be contiguous" % name)
    return t.data_ptr()


def _dp(t):
    """data pointer of an optional tensor the caller already knows to be a contiguous CUDA tensor of the right type"""
    return None if t is None else t.data_ptr()


def _i3(x):
    return _I3(*[int(v) for v in x])


def _ws(nbytes, device):
    return torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------------------
# deferred weight-gradient reduces (csrc/wgrad_reduce.hip)
# --------------------------------------------------------------------------------------
class RsloWgradReduce(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_blocks", C.c_int32), ("ws", C.c_void_p), ("dW", C.c_void_p), ("aux", C.c_void_p),
                ("dbias", C.c_void_p), ("koff", C.c_void_p), ("p", C.c_int32 * 8)]


import threading as _threading
_tls = _threading.local()      # .sink: the ReduceSink installed by THIS thread (the C library's sink is per thread as well)


class ReduceSink:
    """Collects the second stage (partials -> gradient) of the weight-gradient entry points called inside `with
    sink.collect():` instead of launching it per layer; `flush()` runs everything collected as ONE launch on the current
    stream (rslo_wgrad_reduce_many: same block bodies, same bits).  The workspaces / bias partials the descriptors point to are
    kept referenced until the flush.  The caller (rslo_amd/streams.py) flushes before anything can read a gradient."""
    CAP = 256

    def __init__(self):
        self.arr = (RsloWgradReduce * self.CAP)()
        self.count = C.c_int(0)
        self.keep = []

    class _Collect:
        def __init__(self, sink):
            self.sink = sink

        def __enter__(self):
            self.prev = getattr(_tls, "sink", None)
            _tls.sink = self.sink
            _chk(lib().rslo_wgrad_reduce_defer(self.sink.arr, ReduceSink.CAP, C.byref(self.sink.count)), "rslo_wgrad_reduce_defer")
            return self.sink

        def __exit__(self, *exc):
            _tls.sink = self.prev
            if self.prev is not None:
                lib().rslo_wgrad_reduce_defer(self.prev.arr, ReduceSink.CAP, C.byref(self.prev.count))
            else:
                lib().rslo_wgrad_reduce_defer(None, 0, None)
            return False

    def collect(self):
        return ReduceSink._Collect(self)

    def pending(self):
        return self.count.value

    def flush(self):
        n = self.count.value
        if n:
            rc = lib().rslo_wgrad_reduce_many(self.arr, n, _stream())
            self.count.value = 0
            self.keep.clear()
            _chk(rc, "rslo_wgrad_reduce_many")
        else:
            self.keep.clear()


def _keep_for_reduce(*tensors):
    """Inside ReduceSink.collect(): the buffers a deferred reduce will read stay referenced until its flush."""
    sink = getattr(_tls, "sink", None)
    if sink is not None:
        sink.keep.extend(t for t in tensors if t is not None)


# --------------------------------------------------------------------------------------
# voxelization / VFE
# --------------------------------------------------------------------------------------
def voxelize(points, pc_range, voxel_size, grid_xyz, max_points, max_voxels):
    """points [P,F] f32 cuda -> (voxels [max_voxels,T,F], coords [max_voxels,3] zyx, num [max_voxels],
    d_nvox [1] int32 on device).  Rows >= nvox are zero; the caller slices after reading nvox."""
    P, F = points.shape
    dev = points.device
    # one block, laid out voxels | num | coords | nvox: the library zero-fills it with a single memset
    nv, nn, nc = max_voxels * max_points * F, max_voxels, max_voxels * 3
    block = torch.empty((nv + nn + nc + 4,), dtype=torch.int32, device=dev)
    voxels = block[:nv].view(torch.float32).view(max_voxels, max_points, F)
    num = block[nv:nv + nn]
    coords = block[nv + nn:nv + nn + nc].view(max_voxels, 3)
    nvox = block[nv + nn + nc:nv + nn + nc + 1]
    wsb = lib().rslo_voxelize_ws_bytes(P)
    ws = _ws(wsb, dev)
    r = _F6(*[float(v) for v in pc_range])
    v = _F3(*[float(x) for x in voxel_size])
    g = _i3(grid_xyz)
    _chk(lib().rslo_voxelize(_ptr(points, torch.float32, "points"), P, F, r, v, g, int(max_points),
                             int(max_voxels), _ptr(ws), wsb, _ptr(voxels), _ptr(coords), _ptr(num),
                             _ptr(nvox), _stream()), "rslo_voxelize")
    return voxels, coords, num, nvox


def vfe_mean(voxels, num_points):
    M, T, F = voxels.shape
    out = torch.empty((M, F), dtype=torch.float32, device=voxels.device)
    _chk(lib().rslo_vfe_mean(_ptr(voxels, torch.float32, "voxels"), _ptr(num_points, torch.int32, "num_points"),
                             M, T, F, _ptr(out), _stream()), "rslo_vfe_mean")
    return out


# --------------------------------------------------------------------------------------
# site index + rulebooks
# --------------------------------------------------------------------------------------
class SiteIndex:
    """Hash over the active sites of one level: coords [N,4] int32 (b,z,y,x)."""

    def __init__(self, coords, batch, dims):
        self.coords = coords
        self.batch = int(batch)
        self.dims = [int(d) for d in dims]
        N = coords.shape[0]
        self.cap = int(lib().rslo_hash_capacity(N))
        self.keys = torch.empty((self.cap,), dtype=torch.int32, device=coords.device)
        self.vals = torch.empty((self.cap,), dtype=torch.int32, device=coords.device)
        _chk(lib().rslo_hash_build(_ptr(coords, torch.int32, "coords"), N, self.batch, _i3(self.dims),
                                   _ptr(self.keys), _ptr(self.vals), self.cap, _stream()), "rslo_hash_build")


    @classmethod
    def from_parts(cls, coords, batch, dims, keys, vals, cap):
        """A site index whose hash already exists (built inside rslo_plan_encoder)."""
        self = cls.__new__(cls)
        self.coords, self.batch, self.dims = coords, int(batch), [int(d) for d in dims]
        self.keys, self.vals, self.cap = keys, vals, int(cap)
        return self


# -- rslo_plan_encoder: the structures of include/rslo_hip.h -------------------------------------------------------
PLAN_MAX_LEVELS, PLAN_MAX_CLOUDS = 8, 64
PLAN_CNT_OVERFLOW, PLAN_CNT_ROWS, PLAN_CNT_NVOX, PLAN_CNT_BOFF = 0, 1, 32, 176
PLAN_CNT_WORDS = 176 + PLAN_MAX_LEVELS * (PLAN_MAX_CLOUDS + 1)
_L3 = (C.c_int32 * 3) * PLAN_MAX_LEVELS
_LU64 = C.c_uint64 * PLAN_MAX_LEVELS
_LI64 = C.c_int64 * PLAN_MAX_LEVELS


class EncoderSpec(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("dims0", C.c_int32 * 3), ("subm_ks", _L3), ("conv_ks", _L3),
                ("conv_stride", _L3), ("conv_pad", _L3), ("want_pairs", C.c_int32), ("want_orders", C.c_int32),
                ("range6", C.c_float * 6), ("vsize3", C.c_float * 3), ("grid_xyz", C.c_int32 * 3),
                ("max_points", C.c_int32), ("max_voxels", C.c_int32), ("n_features", C.c_int32), ("cap_rows", _LI64)]


class PlanLayout(C.Structure):
    _fields_ = [("total_bytes", C.c_uint64), ("dims", _L3), ("cap_rows", _LI64), ("hash_cap", _LI64),
                ("counts_off", C.c_uint64), ("voxels_off", C.c_uint64), ("num_points_off", C.c_uint64),
                ("coords_frame_off", C.c_uint64), ("coords_off", _LU64), ("keys_off", _LU64), ("vals_off", _LU64),
                ("subm_nbr_off", _LU64), ("subm_pin_off", _LU64), ("subm_pout_off", _LU64), ("subm_koff_off", _LU64),
                ("conv_nbr_off", _LU64), ("conv_nbrT_off", _LU64), ("conv_order_off", _LU64), ("conv_pin_off", _LU64),
                ("conv_pout_off", _LU64), ("conv_koff_off", _LU64), ("scratch_words", C.c_int64),
                ("vox_ws_off", C.c_uint64), ("bitmap_off", C.c_uint64), ("prefix_off", C.c_uint64),
                ("scan_ws_off", C.c_uint64), ("pair_ws_off", C.c_uint64), ("keys_end_off", C.c_uint64),
                ("bitmap_level_off", _LU64), ("bitmap_end_off", C.c_uint64)]


def plan_encoder_layout(spec, n_points):
    """spec: EncoderSpec, n_points: per-cloud point counts -> PlanLayout (offsets into one arena, capacities)."""
    lay = PlanLayout()
    arr = (C.c_int64 * len(n_points))(*[int(n) for n in n_points])
    _chk(lib().rslo_plan_encoder_layout(C.byref(spec), len(n_points), arr, C.byref(lay)), "rslo_plan_encoder_layout")
    return lay


def plan_encoder(spec, lay, clouds, clouds_per_frame, arena, h_counts):
    """Enqueue voxelization + the whole rulebook chain for `clouds` (CUDA fp32 [P,F] tensors, frame-major) on the current
    stream: ONE foreign call, no host read.  arena: uint8 CUDA tensor of >= lay.total_bytes; h_counts: pinned int32
    tensor [PLAN_CNT_WORDS] that receives the counts block (valid once an event recorded after this call has passed)."""
    n = len(clouds)
    ptrs = (C.c_void_p * n)(*[_ptr(c, torch.float32, "cloud") if c.shape[0] else None for c in clouds])
    cnts = (C.c_int64 * n)(*[int(c.shape[0]) for c in clouds])
    _chk(lib().rslo_plan_encoder(C.byref(spec), C.byref(lay), n, int(clouds_per_frame), ptrs, cnts, _ptr(arena),
                                 arena.numel(), C.c_void_p(h_counts.data_ptr()), _stream()), "rslo_plan_encoder")


def plan_encoder_pad_tails(spec, lay, arena):
    """Behind plan_encoder on the current stream: rows past every level's count become padding rows (capacity-sized plan)."""
    _chk(lib().rslo_plan_encoder_pad_tails(C.byref(spec), C.byref(lay), _ptr(arena), _stream()), "rslo_plan_encoder_pad_tails")


def rulebook_subm(index, ks):
    N = index.coords.shape[0]
    K = int(ks[0] * ks[1] * ks[2])
    nbr = torch.empty((N, K), dtype=torch.int32, device=index.coords.device)
    _chk(lib().rslo_rulebook_subm(_ptr(index.coords), N, index.batch, _i3(index.dims), _i3(ks),
                                  _ptr(index.keys), _ptr(index.vals), index.cap, _ptr(nbr), _stream()),
         "rslo_rulebook_subm")
    return nbr


def conv_out_dims(in_dims, ks, stride, pad):
    return [(int(d) + 2 * int(p) - int(k)) // int(s) + 1 for d, k, s, p in zip(in_dims, ks, stride, pad)]


def rulebook_conv(index, ks, stride, pad):
    """Strided conv rulebook.  Returns (out_index: SiteIndex, nbr [M,K], nbrT [N,K]).
    One device->host read (the output count) is unavoidable: it sizes the output tensors."""
    dev = index.coords.device
    N = index.coords.shape[0]
    K = int(ks[0] * ks[1] * ks[2])
    od = conv_out_dims(index.dims, ks, stride, pad)
    words = int(lib().rslo_conv_bitmap_words(index.batch, _i3(od)))
    bitmap = torch.empty((words,), dtype=torch.int32, device=dev)
    prefix = torch.empty((words,), dtype=torch.int32, device=dev)
    swb = lib().rslo_scan_ws_bytes(words)
    sws = _ws(swb, dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    _chk(lib().rslo_conv_out_count(_ptr(index.coords), N, index.batch, _i3(ks), _i3(stride), _i3(pad),
                                   _i3(od), _ptr(bitmap), _ptr(prefix), words, _ptr(sws), swb, _ptr(cnt),
                                   _stream()), "rslo_conv_out_count")
    M = int(cnt.item())
    out_coords = torch.empty((M, 4), dtype=torch.int32, device=dev)
    _chk(lib().rslo_conv_out_coords(_ptr(bitmap), _ptr(prefix), words, index.batch, _i3(od),
                                    _ptr(out_coords), M, _stream()), "rslo_conv_out_coords")
    out_index = SiteIndex(out_coords, index.batch, od)
    nbr = torch.empty((M, K), dtype=torch.int32, device=dev)
    _chk(lib().rslo_rulebook_conv(_ptr(out_coords), M, index.batch, _i3(index.dims), _i3(ks), _i3(stride),
                                  _i3(pad), _ptr(index.keys), _ptr(index.vals), index.cap, _ptr(nbr),
                                  _stream()), "rslo_rulebook_conv")
    nbrT = torch.empty((N, K), dtype=torch.int32, device=dev)
    _chk(lib().rslo_rulebook_conv_T(_ptr(index.coords), N, index.batch, _i3(od), _i3(ks), _i3(stride),
                                    _i3(pad), _ptr(out_index.keys), _ptr(out_index.vals), out_index.cap,
                                    _ptr(nbrT), _stream()), "rslo_rulebook_conv_T")
    return out_index, nbr, nbrT


# --------------------------------------------------------------------------------------
# sparse conv arithmetic
# --------------------------------------------------------------------------------------
SPLIT_BF16 = os.environ.get("RSLO_SPCONV_SPLIT", "1") != "0"     # fp32 via 3-way bf16 splitting on the matrix cores


class WeightSplitDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ws_fwd", C.c_void_p), ("ws_dgrad", C.c_void_p), ("K", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32)]


def weight_split_many(weights):
    """weights: list of [K,cin,cout] fp32 CUDA tensors with 32/64 channels.  -> (plan, [(ws_fwd, ws_dgrad)]); run
    weight_split_run(plan) after every weight update (one launch for all layers and both orientations)."""
    dev = weights[0].device
    sizes = [lib().rslo_weight_split_bytes(*w.shape) for w in weights]
    pool = torch.empty((2 * sum(sizes),), dtype=torch.uint8, device=dev)
    arr = (WeightSplitDesc * len(weights))()
    views, off = [], 0
    for i, (w, n) in enumerate(zip(weights, sizes)):
        f, t = pool[off:off + n], pool[off + n:off + 2 * n]
        off += 2 * n
        views.append((f, t))
        arr[i] = WeightSplitDesc(_ptr(w, torch.float32, "w"), f.data_ptr(), t.data_ptr(), *w.shape)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    return {"table": table, "n": len(weights), "max": max(int(w.numel()) for w in weights), "pool": pool,
            "ptrs": [w.data_ptr() for w in weights]}, views


def weight_split_run(plan):
    _chk(lib().rslo_weight_split_many(_ptr(plan["table"]), plan["n"], plan["max"], _stream()), "rslo_weight_split_many")


def weight_split(W, transpose=False):
    """W [K,Cin,Cout] fp32 -> split-bf16 operand planes for rslo_spconv_fwd_split (transpose: data-gradient operator).
    Operands refreshed for the whole model by weight_split_many (kept on the parameter object) are reused while the
    parameter is unchanged."""
    pre = getattr(W, "_hip_split", None)
    if pre is not None and pre[2] == W._version and pre[3] == W.data_ptr():
        return pre[1] if transpose else pre[0]
    K, cin, cout = W.shape
    cin_op, cout_op = (cout, cin) if transpose else (cin, cout)
    Ws = torch.empty((lib().rslo_weight_split_bytes(K, cin, cout),), dtype=torch.uint8, device=W.device)
    _chk(lib().rslo_weight_split(_ptr(W, torch.float32, "W"), K, cin_op, cout_op, int(transpose), _ptr(Ws), _stream()),
         "rslo_weight_split")
    return Ws


ROW_ORDER = os.environ.get("RSLO_ROW_ORDER", "1") != "0"        # mask-sorted tile order (rslo_rulebook_row_order)


def rulebook_row_order(nbr, flip_k=False):
    """Scheduling order of the table's rows (int32 [n]); None when switched off (RSLO_ROW_ORDER=0)."""
    if not ROW_ORDER:
        return None
    n, K = nbr.shape
    order = torch.empty((n,), dtype=torch.int32, device=nbr.device)
    _chk(lib().rslo_rulebook_row_order(_ptr(nbr, torch.int32, "nbr"), n, K, int(flip_k), _ptr(order), _stream()),
         "rslo_rulebook_row_order")
    return order


def spconv_fwd_split(x, Ws, bias, nbr, cin, cout, flip_k=False, act_slope=1.0, order=None, n_live=None):
    n_out, K = nbr.shape
    if x.shape[1] != cin:
        raise RsloHipError("spconv_fwd_split: shape mismatch")
    out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    if n_live is not None and order is None:       # consumed by the launch that follows
        lib().rslo_spconv_set_live_rows(_ptr(n_live, torch.int32, "n_live"))
    _chk(lib().rslo_spconv_fwd_split(_ptr(x, torch.float32, "x"), cin, _ptr(Ws), _ptr(bias, torch.float32, "bias"),
                                     _ptr(nbr, torch.int32, "nbr"), _ptr(order, torch.int32, "order"), n_out, K, cout,
                                     int(flip_k), float(act_slope), _ptr(out), _stream()), "rslo_spconv_fwd_split")
    return out


def weight_to_bf16(W, transpose=False):
    """W [K,Cin,Cout] fp32 -> bf16 operand plane for rslo_spconv_fwd_bf16."""
    K, cin, cout = W.shape
    cin_op, cout_op = (cout, cin) if transpose else (cin, cout)
    Wb = torch.empty((K * cin * cout,), dtype=torch.bfloat16, device=W.device)
    _chk(lib().rslo_weight_to_bf16(_ptr(W, torch.float32, "W"), K, cin_op, cout_op, int(transpose), _ptr(Wb), _stream()),
         "rslo_weight_to_bf16")
    return Wb


def spconv_fwd_bf16(x, W, bias, nbr, flip_k=False, act_slope=1.0, transpose=False, order=None):
    """bf16 feature path: x [Nin,Cin] bfloat16, W [K,Cin,Cout] fp32 master weights, bias fp32 -> [Nout,Cout] bfloat16.
    transpose=True applies W[k]^T (the data gradient; x is then dout [Nout,Cout], nbr the transposed table)."""
    n_out, K = nbr.shape
    Kw, cin, cout = W.shape
    cin_op, cout_op = (cout, cin) if transpose else (cin, cout)
    if Kw != K or x.shape[1] != cin_op or x.dtype != torch.bfloat16:
        raise RsloHipError("spconv_fwd_bf16: shape / dtype mismatch")
    out = torch.empty((n_out, cout_op), dtype=torch.bfloat16, device=x.device)
    _chk(lib().rslo_spconv_fwd_bf16(_ptr(x, torch.bfloat16, "x"), cin_op, _ptr(weight_to_bf16(W, transpose)),
                                    _ptr(bias, torch.float32, "bias"), _ptr(nbr, torch.int32, "nbr"),
                                    _ptr(order, torch.int32, "order"), n_out, K, cout_op,
                                    int(flip_k), float(act_slope), _ptr(out), _stream()), "rslo_spconv_fwd_bf16")
    return out


def _splittable(cin, cout):
    return SPLIT_BF16 and cin in (32, 64) and cout in (32, 64)


def spconv_fwd(x, W, bias, nbr, flip_k=False, act_slope=1.0, order=None, n_live=None):
    """x [Nin,Cin], W [K,Cin,Cout], nbr [Nout,K] -> [Nout,Cout].  order: optional rulebook_row_order(nbr).
    n_live: device int32 word = how many leading rows of a capacity-laid-out table are real (rslo_spconv_set_live_rows);
    rows past it are left unwritten."""
    n_out, K = nbr.shape
    Kw, cin, cout = W.shape
    if Kw != K or x.shape[1] != cin:
        raise RsloHipError("spconv_fwd: shape mismatch x%s W%s nbr%s" % (tuple(x.shape), tuple(W.shape), tuple(nbr.shape)))
    if _splittable(cin, cout):
        return spconv_fwd_split(x, weight_split(W), bias, nbr, cin, cout, flip_k, act_slope, order, n_live)
    return spconv_fwd_direct(x, W, bias, nbr, flip_k, act_slope, order, n_live)


def spconv_fwd_direct(x, W, bias, nbr, flip_k=False, act_slope=1.0, order=None, n_live=None):
    """rslo_spconv_fwd itself (fp32 MFMA kernels)."""
    n_out, K = nbr.shape
    Kw, cin, cout = W.shape
    out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    if n_live is not None and order is None:       # consumed by the launch that follows
        lib().rslo_spconv_set_live_rows(_ptr(n_live, torch.int32, "n_live"))
    _chk(lib().rslo_spconv_fwd(_ptr(x, torch.float32, "x"), cin, _ptr(W, torch.float32, "W"),
                               _ptr(bias, torch.float32, "bias"), _ptr(nbr, torch.int32, "nbr"),
                               _ptr(order, torch.int32, "order"), n_out, K, cout,
                               int(flip_k), float(act_slope), _ptr(out), _stream()), "rslo_spconv_fwd")
    return out


def weight_transpose(W):
    """W [K,Cin,Cout] -> [K,Cout,Cin]."""
    K, cin, cout = W.shape
    Wt = torch.empty((K, cout, cin), dtype=torch.float32, device=W.device)
    _chk(lib().rslo_weight_transpose(_ptr(W, torch.float32, "W"), K, cin, cout, _ptr(Wt), _stream()),
         "rslo_weight_transpose")
    return Wt


def spconv_dgrad(dout, W, nbrT, flip_k=False, order=None):
    """dout [Nout,Cout], W [K,Cin,Cout], nbrT [Nin,K] -> din [Nin,Cin].
    For MFMA-shaped channel counts the gradient runs through the forward kernel on the transposed weights
    (coalesced weight reads); other shapes use the dedicated entry point."""
    Kw, cin, cout = W.shape
    if _splittable(cout, cin):
        return spconv_fwd_split(dout, weight_split(W, transpose=True), None, nbrT, cout, cin, flip_k, order=order)
    if cin % 16 == 0 and cout % 16 == 0:
        return spconv_fwd(dout, weight_transpose(W), None, nbrT, flip_k=flip_k, order=order)
    return spconv_dgrad_direct(dout, W, nbrT, flip_k, order)


def spconv_dgrad_direct(dout, W, nbrT, flip_k=False, order=None):
    """rslo_spconv_dgrad itself (weights read in place, transposed access)."""
    n_in, K = nbrT.shape
    Kw, cin, cout = W.shape
    if Kw != K or dout.shape[1] != cout:
        raise RsloHipError("spconv_dgrad: shape mismatch")
    din = torch.empty((n_in, cin), dtype=torch.float32, device=dout.device)
    _chk(lib().rslo_spconv_dgrad(_ptr(dout, torch.float32, "dout"), cout, _ptr(W, torch.float32, "W"),
                                 _ptr(nbrT, torch.int32, "nbrT"), _ptr(order, torch.int32, "order"), n_in, K, cin,
                                 int(flip_k), _ptr(din), _stream()), "rslo_spconv_dgrad")
    return din


def spconv_wgrad(x, dout, nbr, cin, cout, with_bias=True):
    n_out, K = nbr.shape
    dev = x.device
    dW = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if with_bias else None
    wsb = lib().rslo_spconv_wgrad_ws_bytes(n_out, K, cin, cout)
    ws = _ws(wsb, dev)
    _chk(lib().rslo_spconv_wgrad(_ptr(x, torch.float32, "x"), cin, _ptr(dout, torch.float32, "dout"), cout,
                                 _ptr(nbr, torch.int32, "nbr"), n_out, K, _ptr(ws), wsb, _ptr(dW), _ptr(db),
                                 _stream()), "rslo_spconv_wgrad")
    return dW, db


def rulebook_pairs(nbr):
    """nbr [rows,K] -> (pairs_in [rows*K cap], pairs_out [rows*K cap], koff [K+1] on device)."""
    n, K = nbr.shape
    dev = nbr.device
    cap = max(n * K, 1)
    pin = torch.empty((cap,), dtype=torch.int32, device=dev)
    pout = torch.empty((cap,), dtype=torch.int32, device=dev)
    koff = torch.empty((K + 1,), dtype=torch.int32, device=dev)
    wsb = lib().rslo_rulebook_pairs_ws_bytes(n, K)
    ws = _ws(wsb, dev)
    _chk(lib().rslo_rulebook_pairs(_ptr(nbr, torch.int32, "nbr"), n, K, _ptr(ws), wsb, _ptr(pin), _ptr(pout),
                                   _ptr(koff), _stream()), "rslo_rulebook_pairs")
    return pin, pout, koff


def spconv_wgrad_pairs(x, dout, pairs, n_out, K, cin, cout, with_bias=True, bias_partial=None):
    """pairs = (pairs_in, pairs_out, koff) over rows of x (in) and dout (out).
    bias_partial: per-block column sums of dout from leaky_bwd(..., colsum=True) (saves a pass over dout)."""
    pin, pout, koff = pairs
    dev = x.device
    dW = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if with_bias else None
    wsb = lib().rslo_spconv_wgrad_pairs_ws_bytes(n_out, K, cin, cout)
    ws = _ws(wsb, dev)
    _chk(lib().rslo_spconv_wgrad_pairs(_ptr(x, torch.float32, "x"), cin, _ptr(dout, torch.float32, "dout"), cout,
                                       _ptr(pin, torch.int32, "pairs_in"), _ptr(pout, torch.int32, "pairs_out"),
                                       _ptr(koff, torch.int32, "koff"), n_out, K, _ptr(ws), wsb, _ptr(dW), _ptr(db),
                                       _ptr(bias_partial if with_bias else None),
                                       0 if bias_partial is None else bias_partial.shape[0], _stream()),
         "rslo_spconv_wgrad_pairs")
    _keep_for_reduce(ws, bias_partial, koff)
    return dW, db


def leaky_bwd(y, dout, slope, colsum=False):
    """g = dout * (y > 0 ? 1 : slope); colsum=True also returns per-block column sums of g ([blocks, cols]) when the
    row width divides 1024 (else None)."""
    g = torch.empty_like(dout)
    if colsum:
        rows, cols = dout.shape
        if cols >= 4 and 1024 % cols == 0:
            nblk = int(lib().rslo_leaky_bwd_colsum_blocks(rows, cols))
            part = torch.empty((nblk, cols), dtype=torch.float32, device=dout.device)
            _chk(lib().rslo_leaky_bwd_colsum(_ptr(y, torch.float32, "y"), _ptr(dout, torch.float32, "dout"), rows, cols,
                                             float(slope), _ptr(g), _ptr(part), _stream()), "rslo_leaky_bwd_colsum")
            return g, part
        colsum = None
    _chk(lib().rslo_leaky_bwd(_ptr(y, torch.float32, "y"), _ptr(dout, torch.float32, "dout"), y.numel(),
                              float(slope), _ptr(g), _stream()), "rslo_leaky_bwd")
    return (g, None) if colsum is None else g


def leaky_bwd_bf16(y, dout, slope, colsum=False):
    """bf16 rows: g = dout * (y > 0 ? 1 : slope) in bf16; colsum -> (g, per-block fp32 column sums [blocks, cols])."""
    rows, cols = dout.shape
    g = torch.empty_like(dout)
    part = None
    if colsum:
        nblk = int(lib().rslo_leaky_bwd_colsum_bf16_blocks(rows, cols))
        part = torch.empty((nblk, cols), dtype=torch.float32, device=dout.device)
    _chk(lib().rslo_leaky_bwd_colsum_bf16(_ptr(y, torch.bfloat16, "y"), _ptr(dout, torch.bfloat16, "dout"), rows, cols,
                                          float(slope), _ptr(g), _ptr(part), _stream()), "rslo_leaky_bwd_colsum_bf16")
    return (g, part) if colsum else g


def spconv_wgrad_pairs_bf16(x, dout, pairs, n_out, K, cin, cout, bias_partial=None):
    """bf16 rows on both sides -> fp32 (dW [K,cin,cout], dbias or None); dbias needs bias_partial (leaky_bwd_bf16)."""
    pin, pout, koff = pairs
    dev = x.device
    dW = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if bias_partial is not None else None
    wsb = lib().rslo_spconv_wgrad_pairs_ws_bytes(n_out, K, cin, cout)
    ws = _ws(wsb, dev)
    _chk(lib().rslo_spconv_wgrad_pairs_bf16(_ptr(x, torch.bfloat16, "x"), cin, _ptr(dout, torch.bfloat16, "dout"), cout,
                                            _ptr(pin, torch.int32, "pairs_in"), _ptr(pout, torch.int32, "pairs_out"),
                                            _ptr(koff, torch.int32, "koff"), n_out, K, _ptr(ws), wsb, _ptr(dW), _ptr(db),
                                            _ptr(bias_partial), 0 if bias_partial is None else bias_partial.shape[0],
                                            _stream()), "rslo_spconv_wgrad_pairs_bf16")
    _keep_for_reduce(ws, bias_partial, koff)
    return dW, db


def segbn_fwd(x, seg_off, S, max_len, gamma, beta, running_mean, running_var, momentum, eps, act_slope):
    """-> (y, save_mean [S,C], save_invstd [S,C]); running stats updated in place, segment after segment."""
    n, Cc = x.shape
    dev = x.device
    y = torch.empty_like(x)
    mean = torch.empty((S, Cc), dtype=torch.float32, device=dev)
    invstd = torch.empty((S, Cc), dtype=torch.float32, device=dev)
    wsb = lib().rslo_segbn_ws_bytes(S, max_len, Cc)
    ws = _ws(wsb, dev)
    _chk(lib().rslo_segbn_fwd(_ptr(x, torch.float32, "x"), Cc, _ptr(seg_off, torch.int32, "seg_off"), S, max_len,
                              _ptr(gamma, torch.float32, "gamma"), _ptr(beta, torch.float32, "beta"),
                              _ptr(running_mean, torch.float32, "running_mean"),
                              _ptr(running_var, torch.float32, "running_var"), float(momentum), float(eps),
                              float(act_slope), _ptr(ws), wsb, _ptr(y), _ptr(mean), _ptr(invstd), _stream()),
         "rslo_segbn_fwd")
    return y, mean, invstd


def bn1d_eval_act(x, running_mean, running_var, gamma, beta, eps, act_slope=1.0, n_live=None):
    """Eval-mode BatchNorm1d + (Leaky)ReLU over [n, C] fp32 rows in one launch (rslo_bn1d_eval_act)."""
    n, Cc = x.shape
    y = torch.empty_like(x)
    _chk(lib().rslo_bn1d_eval_act(_ptr(x, torch.float32, "x"), n, Cc, _ptr(running_mean, torch.float32, "running_mean"),
                                  _ptr(running_var, torch.float32, "running_var"), _ptr(gamma, torch.float32, "gamma"),
                                  _ptr(beta, torch.float32, "beta"), float(eps), float(act_slope),
                                  _ptr(n_live, torch.int32, "n_live"), _ptr(y), _stream()), "rslo_bn1d_eval_act")
    return y


def segbn_bwd(x, y, gy, seg_off, S, max_len, gamma, mean, invstd, act_slope):
    n, Cc = x.shape
    dev = x.device
    gx = torch.empty_like(x)
    dgamma = torch.empty((Cc,), dtype=torch.float32, device=dev)
    dbeta = torch.empty((Cc,), dtype=torch.float32, device=dev)
    wsb = lib().rslo_segbn_ws_bytes(S, max_len, Cc) + 2 * S * Cc * 4
    ws = _ws(wsb, dev)
    _chk(lib().rslo_segbn_bwd(_ptr(x, torch.float32, "x"), _ptr(y, torch.float32, "y"), _ptr(gy, torch.float32, "gy"),
                              Cc, _ptr(seg_off, torch.int32, "seg_off"), S, max_len,
                              _ptr(gamma, torch.float32, "gamma"), _ptr(mean), _ptr(invstd), float(act_slope),
                              _ptr(ws), wsb, _ptr(gx), _ptr(dgamma), _ptr(dbeta), _stream()), "rslo_segbn_bwd")
    return gx, dgamma, dbeta


def dense_scatter(feat, coords, batch, dims, frames=1, out=None):
    """[M,C] rows -> [batch, C, D, H, W]; frames > 1: [batch / frames, frames, C, D, H, W] (rows of frame t of sample b
    carry the batch index t * (batch / frames) + b).  out: a contiguous fp32 buffer of that many elements to write into
    (the static input of a replayed head graph: rslo_amd/headgraph.py) -- returned viewed to the shape above."""
    M, Cc = feat.shape
    shape = (batch, Cc, dims[0], dims[1], dims[2]) if frames == 1 else (batch // frames, frames, Cc, dims[0], dims[1], dims[2])
    if out is not None:
        n = 1
        for v in shape:
            n *= int(v)
        if not (out.is_contiguous() and out.dtype == torch.float32 and out.device == feat.device and out.numel() == n):
            raise RsloHipError("dense_scatter: `out` must be a contiguous fp32 buffer of %d elements on %s" % (n, feat.device))
        # a tensor of its own over the same memory (not a view of `out`: the result is an autograd output)
        strides, acc = [], 1
        for v in reversed(shape):
            strides.append(acc)
            acc *= int(v)
        out = torch.empty(0, dtype=torch.float32, device=feat.device).set_(out.untyped_storage(), out.storage_offset(), shape,
                                                                          tuple(reversed(strides)))
    else:
        out = torch.empty(shape, dtype=torch.float32, device=feat.device)
    _chk(lib().rslo_dense_scatter_frames(_ptr(feat, torch.float32, "feat"), _ptr(coords, torch.int32, "coords"), M, Cc,
                                         int(batch), int(frames), _i3(dims), _ptr(out), _stream()), "rslo_dense_scatter")
    return out


def dense_gather(dense, coords, C_, batch, dims, frames=1):
    M = coords.shape[0]
    out = torch.empty((M, C_), dtype=torch.float32, device=dense.device)
    _chk(lib().rslo_dense_gather_frames(_ptr(dense, torch.float32, "dense"), _ptr(coords, torch.int32, "coords"), M, C_,
                                        int(batch), int(frames), _i3(dims), _ptr(out), _stream()), "rslo_dense_gather")
    return out


def bev_channel_sums(bev, groups, masks=False):
    """bev [B, groups * Cg, H, W] fp32 -> [B, groups, H, W]: per-cell sum over each channel group, one pass.
    masks=True: also (mask_f float, mask_b bool, outside_b bool) [B,1,H,W] = occupancy of group 0 from the same launch."""
    B, Ct, H, W = bev.shape
    out = torch.empty((B, groups, H, W), dtype=torch.float32, device=bev.device)
    if masks:
        mf = torch.empty((B, 1, H, W), dtype=torch.float32, device=bev.device)
        mb = torch.empty((B, 1, H, W), dtype=torch.bool, device=bev.device)
        ob = torch.empty((B, 1, H, W), dtype=torch.bool, device=bev.device)
        _chk(lib().rslo_bev_channel_sums_masks(_ptr(bev, torch.float32, "bev"), B, int(groups), Ct // int(groups), H * W,
                                               _ptr(out), _ptr(mf), _ptr(mb), _ptr(ob), _stream()),
             "rslo_bev_channel_sums_masks")
        return out, mf, mb, ob
    _chk(lib().rslo_bev_channel_sums(_ptr(bev, torch.float32, "bev"), B, int(groups), Ct // int(groups), H * W, _ptr(out),
                                     _stream()), "rslo_bev_channel_sums")
    return out


def pose_tail_fwd(odom):
    """odom [B,7] -> (t [B,3], r [B,4] = q / (|q| + 1e-12))."""
    B = odom.shape[0]
    t = torch.empty((B, 3), dtype=torch.float32, device=odom.device)
    r = torch.empty((B, 4), dtype=torch.float32, device=odom.device)
    _chk(lib().rslo_pose_tail_fwd(_ptr(odom, torch.float32, "odom"), B, _ptr(t), _ptr(r), _stream()), "rslo_pose_tail_fwd")
    return t, r


def pose_tail_bwd(odom, g_t, g_r):
    B = odom.shape[0]
    d = torch.empty((B, 7), dtype=torch.float32, device=odom.device)
    _chk(lib().rslo_pose_tail_bwd(_ptr(odom, torch.float32, "odom"), _dp(g_t), _dp(g_r), B, _ptr(d), _stream()),
         "rslo_pose_tail_bwd")
    return d


def cat_upsample_fwd(a, b, scale):
    """a [B,Ca,H,W], b [B,Cb,H,W] contiguous fp32 -> nearest upsampling by `scale` of cat([a, b], 1), one launch."""
    B, Ca, H, W = a.shape
    Cb = b.shape[1]
    if tuple(b.shape) != (B, Cb, H, W):
        raise ValueError("cat_upsample_fwd: shapes %s / %s" % (tuple(a.shape), tuple(b.shape)))
    out = torch.empty((B, Ca + Cb, H * scale, W * scale), dtype=torch.float32, device=a.device)
    _chk(lib().rslo_cat_upsample_fwd(_ptr(a, torch.float32, "a"), _ptr(b, torch.float32, "b"), B, Ca, Cb, H, W, int(scale),
                                     _ptr(out), _stream()), "rslo_cat_upsample_fwd")
    return out


def cat_upsample_bwd(grad, Ca, Cb, scale, need_a=True, need_b=True):
    """grad [B,Ca+Cb,s*H,s*W] -> (da [B,Ca,H,W], db [B,Cb,H,W]) (None where not needed)."""
    B, Ct, Hs, Ws = grad.shape
    H, W = Hs // scale, Ws // scale
    da = torch.empty((B, Ca, H, W), dtype=torch.float32, device=grad.device) if need_a else None
    db = torch.empty((B, Cb, H, W), dtype=torch.float32, device=grad.device) if need_b else None
    _chk(lib().rslo_cat_upsample_bwd(_ptr(grad, torch.float32, "grad"), B, Ca, Cb, H, W, int(scale), _dp(da), _dp(db),
                                     _stream()), "rslo_cat_upsample_bwd")
    return da, db


def bev_display(sums, channels):
    """sums [B, T, H, W] from bev_channel_sums over `channels` channels per frame -> (feature_mask [B,1,H,W],
    [middle_feature_t [B,1,H,W] for t]) in two launches (see rslo_bev_display)."""
    B, T, H, W = sums.shape
    mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=sums.device)
    disp = torch.empty((T, B, 1, H, W), dtype=torch.float32, device=sums.device)
    wsb = lib().rslo_bev_display_ws_bytes(T)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=sums.device)
    _chk(lib().rslo_bev_display(_ptr(sums, torch.float32, "sums"), B, T, int(channels), H * W, _ptr(mask), _ptr(disp),
                                ws.data_ptr(), wsb, _stream()), "rslo_bev_display")
    return mask, list(disp.unbind(0))


# --------------------------------------------------------------------------------------
# chamfer
# --------------------------------------------------------------------------------------
def chamfer_nn(xyz1, xyz2, dist=None, idx=None, ncnt=None, mcnt=None, method=None):
    """ncnt / mcnt: optional int32 [B] device tensors for a ragged (padded) batch.
    method: None (library picks), "brute" (exhaustive scan) or "grid" (pruned search) -- identical results."""
    entry = {None: "rslo_chamfer_nn_ragged", "brute": "rslo_chamfer_brute_nn", "grid": "rslo_chamfer_grid_nn"}[method]
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    dev = xyz1.device
    if dist is None:
        dist = torch.empty((B, N), dtype=torch.float32, device=dev)
    if idx is None:
        idx = torch.empty((B, N), dtype=torch.int32, device=dev)
    wsb = lib().rslo_chamfer_ws_bytes(B, N, M)
    ws = _ws(wsb, dev)
    _chk(getattr(lib(), entry)(_ptr(xyz1, torch.float32, "xyz1"), _ptr(xyz2, torch.float32, "xyz2"), B, N, M,
                               _ptr(ncnt, torch.int32, "ncnt"), _ptr(mcnt, torch.int32, "mcnt"),
                               _ptr(dist, torch.float32, "dist"), _ptr(idx, torch.int32, "idx"), _ptr(ws), wsb,
                               _stream()), entry)
    return dist, idx


def chamfer_grad(xyz1, xyz2, graddist1, idx1, g1=None, g2=None):
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    if g1 is None:
        g1 = torch.empty_like(xyz1)
    if g2 is None:
        g2 = torch.empty_like(xyz2)
    _chk(lib().rslo_chamfer_grad(_ptr(xyz1, torch.float32, "xyz1"), _ptr(xyz2, torch.float32, "xyz2"), B, N, M,
                                 _ptr(graddist1, torch.float32, "graddist1"), _ptr(idx1, torch.int32, "idx1"),
                                 _ptr(g1), _ptr(g2), _stream()), "rslo_chamfer_grad")
    return g1, g2


# --------------------------------------------------------------------------------------
# consistency loss
# --------------------------------------------------------------------------------------
def cov_residual_fwd(p1, tgt, cov1, cov2, idx, dist, thr, Rd, reg_weight):
    """-> loss [B], cnt [B] (see include/rslo_hip.h)."""
    B, N, _ = p1.shape
    M = tgt.shape[1]
    dev = p1.device
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.float32, device=dev)
    wsb = lib().rslo_cov_residual_ws_bytes(B, N)
    ws = _ws(wsb, dev)
    _chk(lib().rslo_cov_residual_fwd(_ptr(p1, torch.float32, "p1"), _ptr(tgt, torch.float32, "tgt"),
                                     _ptr(cov1, torch.float32, "cov1"), _ptr(cov2, torch.float32, "cov2"),
                                     _ptr(idx, torch.int32, "idx"), _ptr(dist, torch.float32, "dist"),
                                     _ptr(thr, torch.float32, "thr"), _ptr(Rd, torch.float32, "Rd"), B, N, M,
                                     float(reg_weight), _ptr(ws), wsb, _ptr(loss), _ptr(cnt), _stream()),
         "rslo_cov_residual_fwd")
    return loss, cnt


_resid_bwd_ws = {}


def cov_residual_bwd(p1, tgt, cov1, cov2, idx, dist, thr, Rd, gloss, cnt, reg_weight, need_gp1=False):
    B, N, _ = p1.shape
    M = tgt.shape[1]
    gp1 = torch.empty_like(p1) if need_gp1 else None
    gtgt = torch.empty_like(tgt)
    gcov1 = torch.empty_like(cov1)
    gcov2 = torch.empty_like(cov2)
    key = (B, N, M)
    wsb = _resid_bwd_ws.get(key)
    if wsb is None:
        wsb = _resid_bwd_ws[key] = lib().rslo_cov_residual_bwd_ws_bytes(B, N, M)
    ws = _ws(wsb, p1.device)
    _chk(lib().rslo_cov_residual_bwd(_ptr(p1, torch.float32, "p1"), _ptr(tgt, torch.float32, "tgt"),
                                     _ptr(cov1, torch.float32, "cov1"), _ptr(cov2, torch.float32, "cov2"),
                                     _ptr(idx, torch.int32, "idx"), _ptr(dist, torch.float32, "dist"),
                                     _ptr(thr, torch.float32, "thr"), _ptr(Rd, torch.float32, "Rd"),
                                     _ptr(gloss, torch.float32, "gloss"), _ptr(cnt, torch.float32, "cnt"), B, N, M,
                                     float(reg_weight), _ptr(gp1), _ptr(gtgt), _ptr(gcov1), _ptr(gcov2), _ptr(ws), wsb,
                                     _stream()),
         "rslo_cov_residual_bwd")
    return gp1, gtgt, gcov1, gcov2


def icp_step(p1, n1, tgt, idx, dist, thr, res_r, res_t, first=False):
    """One Kabsch refinement over the ROI; composes res_r [B,3,3] / res_t [B,3] IN PLACE.
    first: the running motion is the identity -- res_r / res_t are written without being read (no fill by the caller)."""
    B, N, _ = p1.shape
    M = tgt.shape[1]
    wsb = lib().rslo_icp_ws_bytes(B, N)
    ws = _ws(wsb, p1.device)
    entry = "rslo_icp_step_first" if first else "rslo_icp_step"
    _chk(getattr(lib(), entry)(_ptr(p1, torch.float32, "p1"), _ptr(n1, torch.float32, "n1"),
                               _ptr(tgt, torch.float32, "tgt"), _ptr(idx, torch.int32, "idx"),
                               _ptr(dist, torch.float32, "dist"), _ptr(thr, torch.float32, "thr"), B, N, M, _ptr(ws), wsb,
                               _ptr(res_r, torch.float32, "res_r"), _ptr(res_t, torch.float32, "res_t"), None, None,
                               _stream()), entry)


def transform_points(x, R, t):
    B, M, _ = x.shape
    out = torch.empty_like(x)
    _chk(lib().rslo_transform_points(_ptr(x, torch.float32, "x"), _ptr(R, torch.float32, "R"),
                                     _ptr(t, torch.float32, "t"), B, M, _ptr(out), _stream()), "rslo_transform_points")
    return out


# --------------------------------------------------------------------------------------
# pyramid supervision
# --------------------------------------------------------------------------------------
class _PyramidLevel(C.Structure):
    _fields_ = [("pred", C.c_void_p), ("mask", C.c_void_p), ("dpred", C.c_void_p),
                ("h", C.c_int32), ("w", C.c_int32), ("mask_channels", C.c_int32)]


_py_done = {}


def _pyramid_levels(preds, masks, dpreds=None):
    B = preds[0].shape[0]
    arr = (_PyramidLevel * len(preds))()
    for l, (p, m) in enumerate(zip(preds, masks)):
        if p.dim() != 4 or p.shape[1] != 7 or m.shape[0] != B or p.shape[0] != B or m.shape[2:] != p.shape[2:]:
            raise RsloHipError("pyramid_l2: level %d pred %s / mask %s" % (l, tuple(p.shape), tuple(m.shape)))
        arr[l].pred = _ptr(p, torch.float32, "pred")
        arr[l].mask = _ptr(m, torch.float32, "mask")
        arr[l].dpred = _ptr(dpreds[l], torch.float32, "dpred") if dpreds is not None else None
        arr[l].h, arr[l].w, arr[l].mask_channels = p.shape[2], p.shape[3], m.shape[1]
    return arr, B


def pyramid_l2_fwd(preds, masks, tq, H0, W0, origin, vsize):
    """preds[l] [B,7,h,w], masks[l] [B,Cm,h,w], tq [B,7] -> (loss_b [L,B,2], den [L,B,2])."""
    arr, B = _pyramid_levels(preds, masks)
    dev = tq.device
    L = len(preds)
    done = _py_done.get(dev)
    if done is None or done.numel() < L * B:
        done = _py_done[dev] = torch.zeros((max(L * B, 64),), dtype=torch.int32, device=dev)
    wsb = lib().rslo_pyramid_l2_ws_bytes(arr, L, B)
    ws = _ws(wsb, dev)
    loss_b = torch.empty((L, B, 2), dtype=torch.float32, device=dev)
    den = torch.empty((L, B, 2), dtype=torch.float32, device=dev)
    _chk(lib().rslo_pyramid_l2_fwd(arr, L, B, _ptr(tq, torch.float32, "tq"), int(H0), int(W0),
                                   _F3(*[float(v) for v in origin]), _F3(*[float(v) for v in vsize]), _ptr(ws), wsb,
                                   _ptr(done), _ptr(loss_b), _ptr(den), _stream()), "rslo_pyramid_l2_fwd")
    return loss_b, den


def pyramid_l2_bwd(preds, masks, tq, H0, W0, origin, vsize, grad_loss_b, den):
    dpreds = [torch.empty_like(p) for p in preds]
    arr, B = _pyramid_levels(preds, masks, dpreds)
    _chk(lib().rslo_pyramid_l2_bwd(arr, len(preds), B, _ptr(tq, torch.float32, "tq"), int(H0), int(W0),
                                   _F3(*[float(v) for v in origin]), _F3(*[float(v) for v in vsize]),
                                   _ptr(grad_loss_b, torch.float32, "grad"), _ptr(den, torch.float32, "den"),
                                   _stream()), "rslo_pyramid_l2_bwd")
    return dpreds


class HeadMasks(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("conf", C.c_void_p), ("tq", C.c_void_p), ("tq_g", C.c_void_p),
                ("pred", C.c_void_p * 3), ("w", C.c_void_p * 4), ("occ", C.c_void_p * 4), ("mpred", C.c_void_p * 3),
                ("mtq", C.c_void_p), ("mtq_g", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("levels", C.c_int32)]


class HeadMasksBwd(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("occ", C.c_void_p * 4), ("g_mpred", C.c_void_p * 3), ("g_mtq", C.c_void_p),
                ("d_pred", C.c_void_p * 3), ("d_tq", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("levels", C.c_int32)]


def tq_normalize_fwd(tq):
    B, _, H, W = tq.shape
    out = torch.empty_like(tq)
    _chk(lib().rslo_tq_normalize_fwd(_ptr(tq, torch.float32, "tq"), B, H * W, _ptr(out), _stream()), "rslo_tq_normalize_fwd")
    return out


def tq_normalize_bwd(tq, g):
    B, _, H, W = tq.shape
    d = torch.empty_like(tq)
    _chk(lib().rslo_tq_normalize_bwd(_ptr(tq, torch.float32, "tq"), _ptr(g, torch.float32, "grad"), B, H * W, _ptr(d),
                                     _stream()), "rslo_tq_normalize_bwd")
    return d


def conf_softmax_fwd(t_logit, r_logit, outside, temperature):
    """t_logit, r_logit [B,1,H,W] fp32, outside [B,1,H,W] bool -> (t_conf, r_conf [B,1,H,W], conf_temp [B,2,H,W])."""
    B, _, H, W = t_logit.shape
    t_conf, r_conf = torch.empty_like(t_logit), torch.empty_like(r_logit)
    ct = torch.empty((B, 2, H, W), dtype=torch.float32, device=t_logit.device)
    _chk(lib().rslo_conf_softmax_fwd(_ptr(t_logit, torch.float32, "t_logit"), _ptr(r_logit, torch.float32, "r_logit"),
                                     _ptr(outside, torch.bool, "outside"), B, H * W, float(temperature), _ptr(t_conf),
                                     _ptr(r_conf), _ptr(ct), _stream()), "rslo_conf_softmax_fwd")
    return t_conf, r_conf, ct


def conf_softmax_bwd(t_conf, r_conf, g_t, g_r, outside):
    B, _, H, W = t_conf.shape
    d_t, d_r = torch.empty_like(t_conf), torch.empty_like(r_conf)
    _chk(lib().rslo_conf_softmax_bwd(_ptr(t_conf, torch.float32, "t_conf"), _ptr(r_conf, torch.float32, "r_conf"),
                                     _ptr(g_t, torch.float32, "g_t"), _ptr(g_r, torch.float32, "g_r"),
                                     _ptr(outside, torch.bool, "outside"), B, H * W, _ptr(d_t), _ptr(d_r), _stream()),
         "rslo_conf_softmax_bwd")
    return d_t, d_r


def head_masks_fwd(mask, conf, tq, tq_g, preds):
    """mask [B,1,H,W], conf [B,2,H,W], tq / tq_g [B,7,H,W], preds = [pred of level 1, level 2, ...] (finest first)
    -> (w [levels], occ [levels] (occ[0] = mask), mpreds, mtq, mtq_g)."""
    B, _, H, W = mask.shape
    dev = mask.device
    L = 1 + len(preds)
    a = HeadMasks()
    a.mask, a.conf, a.tq, a.tq_g = _ptr(mask, torch.float32, "mask"), _ptr(conf, torch.float32, "conf"), \
        _ptr(tq, torch.float32, "tq"), _ptr(tq_g, torch.float32, "tq_g")
    w, occ, mp = [], [mask], []
    for k in range(L):
        wk = torch.empty((B, 2, H >> k, W >> k), dtype=torch.float32, device=dev)
        w.append(wk)
        a.w[k] = wk.data_ptr()
        if k:
            ok = torch.empty((B, 1, H >> k, W >> k), dtype=torch.float32, device=dev)
            occ.append(ok)
            a.occ[k] = ok.data_ptr()
            a.pred[k - 1] = _ptr(preds[k - 1], torch.float32, "pred")
            m = torch.empty_like(preds[k - 1])
            mp.append(m)
            a.mpred[k - 1] = m.data_ptr()
    mtq, mtq_g = torch.empty_like(tq), torch.empty_like(tq_g)
    a.mtq, a.mtq_g = mtq.data_ptr(), mtq_g.data_ptr()
    a.B, a.H, a.W, a.levels = B, H, W, L
    _chk(lib().rslo_head_masks_fwd(C.byref(a), _stream()), "rslo_head_masks_fwd")
    return w, occ, mp, mtq, mtq_g


def head_masks_bwd(mask, occ, g_mpreds, g_mtq, pred_shapes):
    B, _, H, W = mask.shape
    dev = mask.device
    L = len(occ)
    a = HeadMasksBwd()
    a.mask = mask.data_ptr()
    a.g_mtq = _dp(g_mtq)
    d_tq = torch.empty((B, 7, H, W), dtype=torch.float32, device=dev)
    a.d_tq = d_tq.data_ptr()
    d_preds = []
    for k in range(1, L):
        a.occ[k] = occ[k].data_ptr()
        a.g_mpred[k - 1] = _dp(g_mpreds[k - 1])
        d = torch.empty(pred_shapes[k - 1], dtype=torch.float32, device=dev)
        d_preds.append(d)
        a.d_pred[k - 1] = d.data_ptr()
    a.B, a.H, a.W, a.levels = B, H, W, L
    _chk(lib().rslo_head_masks_bwd(C.byref(a), _stream()), "rslo_head_masks_bwd")
    return d_preds, d_tq


class LossTail(C.Structure):
    _fields_ = [("t_pred", C.c_void_p), ("t_tgt", C.c_void_p), ("q_pred", C.c_void_p), ("q_tgt", C.c_void_p),
                ("alpha_T", C.c_void_p), ("alpha_R", C.c_void_p), ("pyr_loss_b", C.c_void_p), ("alpha_pT", C.c_void_p),
                ("alpha_pR", C.c_void_p), ("pair_loss", C.c_void_p), ("alpha_C", C.c_void_p), ("B", C.c_int32),
                ("L", C.c_int32), ("n_pairs", C.c_int32), ("reserved", C.c_int32), ("w_T", C.c_float), ("w_R", C.c_float),
                ("w_pT", C.c_float), ("w_pR", C.c_float), ("c_scale", C.c_float), ("level_w", C.c_float * 8)]


def loss_tail_fwd(desc, device):
    out = torch.empty((5,), dtype=torch.float32, device=device)
    _chk(lib().rslo_loss_tail_fwd(C.byref(desc), _ptr(out), _stream()), "rslo_loss_tail_fwd")
    return out


def loss_tail_bwd(desc, grad_out, B, L, n_pairs):
    dev = grad_out.device
    d_t = torch.empty((B, 3), dtype=torch.float32, device=dev)
    d_q = torch.empty((B, 4), dtype=torch.float32, device=dev)
    d_pyr = torch.empty((L, B, 2), dtype=torch.float32, device=dev) if L else None
    d_pair = torch.empty((n_pairs,), dtype=torch.float32, device=dev) if n_pairs else None
    d_alpha = torch.empty((5, 4), dtype=torch.float32, device=dev)      # 16-byte slots (include/rslo_hip.h: stride 4)
    _chk(lib().rslo_loss_tail_bwd(C.byref(desc), _ptr(grad_out, torch.float32, "grad"), _ptr(d_t), _ptr(d_q), _ptr(d_pyr),
                                  _ptr(d_pair), _ptr(d_alpha), _stream()), "rslo_loss_tail_bwd")
    return d_t, d_q, d_pyr, d_pair, d_alpha[:, 0]


def pad_rows_fwd(src, off, length, Lmax):
    """src [N,C] -> [B,Lmax,C]: rows off[b] .. off[b]+len[b] of src, zero padded."""
    N, Cc = src.shape
    B = off.shape[0]
    out = torch.empty((B, Lmax, Cc), dtype=torch.float32, device=src.device)
    _chk(lib().rslo_pad_rows_fwd(_ptr(src, torch.float32, "src"), N, Cc, _ptr(off, torch.int32, "off"),
                                 _ptr(length, torch.int32, "len"), B, int(Lmax), _ptr(out), _stream()), "rslo_pad_rows_fwd")
    return out


def pair_rows_fwd(feats, conf, off, length, Lmax):
    """feats [N,F] (F = 7: xyz, intensity, normal | 6: xyz, normal), conf [N,Cc] -> xyz [B,Lmax,3], nrm [B,Lmax,3],
    cov [B,Lmax,Cc]: rows off[b] .. off[b]+len[b] of both, zero padded, in one launch."""
    N, F = feats.shape
    Cc = conf.shape[1]
    B = off.shape[0]
    dev = feats.device
    xyz = torch.empty((B, Lmax, 3), dtype=torch.float32, device=dev)
    nrm = torch.empty((B, Lmax, 3), dtype=torch.float32, device=dev)
    cov = torch.empty((B, Lmax, Cc), dtype=torch.float32, device=dev)
    _chk(lib().rslo_pair_rows_fwd(_ptr(feats, torch.float32, "feats"), N, F, _ptr(conf, torch.float32, "conf"), Cc,
                                  _ptr(off, torch.int32, "off"), _ptr(length, torch.int32, "len"), B, int(Lmax),
                                  _ptr(xyz), _ptr(nrm), _ptr(cov), _stream()), "rslo_pair_rows_fwd")
    return xyz, nrm, cov


def pad_rows_bwd(dout, off, length, N):
    B, Lmax, Cc = dout.shape
    dsrc = torch.empty((N, Cc), dtype=torch.float32, device=dout.device)
    _chk(lib().rslo_pad_rows_bwd(_ptr(dout, torch.float32, "dout"), N, Cc, _ptr(off, torch.int32, "off"),
                                 _ptr(length, torch.int32, "len"), B, int(Lmax), _ptr(dsrc), _stream()), "rslo_pad_rows_bwd")
    return dsrc


def _rows_view(x):
    """[B,M,3] view whose rows are `stride` floats apart inside a contiguous [B,M,C] tensor -> (ptr, stride)."""
    B, M, three = x.shape
    if three != 3 or x.dtype != torch.float32 or not x.is_cuda:
        raise RsloHipError("transform_rows: need a float32 cuda [B,M,3] tensor")
    sb, sm, sc = x.stride()
    if sc != 1 or sb != M * sm or sm < 3:
        x = x.contiguous()
        sm = 3
    return x, C.c_void_p(x.data_ptr()), sm


def transform_rows(x, R, t=None):
    """out[b,j] = R[b] x[b,j] (+ t[b]);  x may be a column slice of a wider row-major tensor."""
    x, ptr, stride = _rows_view(x)
    B, M, _ = x.shape
    out = torch.empty((B, M, 3), dtype=torch.float32, device=x.device)
    _chk(lib().rslo_transform_rows(ptr, stride, _ptr(R, torch.float32, "R"), _ptr(t, torch.float32, "t"), B, M,
                                   _ptr(out), _stream()), "rslo_transform_rows")
    return out


_tr_done = {}


def transform_rows_bwd(x, gout):
    """-> (dR [B,3,3], dt [B,3]) of out = R x + t."""
    x, ptr, stride = _rows_view(x)
    B, M, _ = x.shape
    dev = x.device
    done = _tr_done.get(dev)
    if done is None or done.numel() < B:
        done = _tr_done[dev] = torch.zeros((max(B, 64),), dtype=torch.int32, device=dev)
    wsb = lib().rslo_transform_rows_bwd_ws_bytes(B, M)
    ws = _ws(wsb, dev)
    dR = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    dt = torch.empty((B, 3), dtype=torch.float32, device=dev)
    _chk(lib().rslo_transform_rows_bwd(ptr, stride, _ptr(gout, torch.float32, "gout"), B, M, _ptr(ws), wsb, _ptr(done),
                                       _ptr(dR), _ptr(dt), _stream()), "rslo_transform_rows_bwd")
    return dR, dt


# --------------------------------------------------------------------------------------
# fused (Sync)BatchNorm2d + activation + residual of the dense head
# --------------------------------------------------------------------------------------
_bn_done = {}


def _bn_counters(dev, C_):
    done = _bn_done.get(dev)
    if done is None or done.numel() < C_:
        done = _bn_done[dev] = torch.zeros((max(C_, 512),), dtype=torch.int32, device=dev)
    return done


def bn2d_stats(x):
    """x [N,C,H,W] -> stats [2C+1] float64 (sum, sum of squares per channel, element count)."""
    N, Cc, H, W = x.shape
    dev = x.device
    wsb = lib().rslo_bn2d_ws_bytes(N, Cc, H * W)
    ws = _ws(wsb, dev)
    stats = torch.empty((2 * Cc + 1,), dtype=torch.float64, device=dev)
    _chk(lib().rslo_bn2d_stats(_ptr(x, torch.float32, "x"), N, Cc, H * W, _ptr(ws), wsb, _ptr(_bn_counters(dev, Cc)),
                               _ptr(stats), _stream()), "rslo_bn2d_stats")
    return stats


def bn2d_apply(x, res, stats, gamma, beta, running_mean, running_var, momentum, eps, slope):
    N, Cc, H, W = x.shape
    dev = x.device
    y = torch.empty_like(x)
    mean = torch.empty((Cc,), dtype=torch.float32, device=dev)
    invstd = torch.empty((Cc,), dtype=torch.float32, device=dev)
    _chk(lib().rslo_bn2d_apply(_ptr(x, torch.float32, "x"), _ptr(res, torch.float32, "res"), _ptr(stats, torch.float64),
                               _ptr(gamma, torch.float32, "gamma"), _ptr(beta, torch.float32, "beta"), N, Cc, H * W,
                               float(eps), float(momentum), float(slope), _ptr(running_mean, torch.float32),
                               _ptr(running_var, torch.float32), _ptr(mean), _ptr(invstd), _ptr(y), _stream()),
         "rslo_bn2d_apply")
    return y, mean, invstd


def bn2d_bwd_reduce(dy, y, x, mean, invstd, slope, has_act, want_affine=True):
    N, Cc, H, W = x.shape
    dev = x.device
    wsb = lib().rslo_bn2d_ws_bytes(N, Cc, H * W)
    ws = _ws(wsb, dev)
    red = torch.empty((2 * Cc,), dtype=torch.float64, device=dev)
    dgamma = torch.empty((Cc,), dtype=torch.float32, device=dev) if want_affine else None
    dbeta = torch.empty((Cc,), dtype=torch.float32, device=dev) if want_affine else None
    _chk(lib().rslo_bn2d_bwd_reduce(_ptr(dy, torch.float32, "dy"), _ptr(y, torch.float32, "y"), _ptr(x, torch.float32, "x"),
                                    _ptr(mean), _ptr(invstd), N, Cc, H * W, float(slope), int(has_act), _ptr(ws), wsb,
                                    _ptr(_bn_counters(dev, Cc)), _ptr(red), _ptr(dgamma), _ptr(dbeta), _stream()),
         "rslo_bn2d_bwd_reduce")
    return red, dgamma, dbeta


def bn2d_bwd_apply(dy, y, x, gamma, mean, invstd, red, count, slope, has_act, want_res, count_dev=None):
    N, Cc, H, W = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_res else None
    _chk(lib().rslo_bn2d_bwd_apply(_ptr(dy, torch.float32, "dy"), _ptr(y, torch.float32, "y"), _ptr(x, torch.float32, "x"),
                                   _ptr(gamma, torch.float32, "gamma"), _ptr(mean), _ptr(invstd), _ptr(red, torch.float64),
                                   float(count), _ptr(count_dev, torch.float64, "count_dev"), N, Cc, H * W, float(slope),
                                   int(has_act), _ptr(dx), _ptr(dres),
                                   _stream()), "rslo_bn2d_bwd_apply")
    return dx, dres


_bn_ws_bytes = {}


def _bn2d_ws(N, Cc, HW, dev):
    key = (N, Cc, HW)
    n = _bn_ws_bytes.get(key)
    if n is None:
        n = _bn_ws_bytes[key] = lib().rslo_bn2d_ws_bytes(N, Cc, HW)
    return torch.empty((n,), dtype=torch.uint8, device=dev), n


def bn2d_fwd_local(x, res, gamma, beta, running_mean, running_var, momentum, eps, slope):
    """Single-rank y = act(BN_train(x) + res): two launches.  -> y, save_mean, save_invstd.  (Hot host path: x / res are
    checked, everything else comes from module state or is allocated here.)"""
    N, Cc, H, W = x.shape
    dev = x.device
    HW = H * W
    ws, wsb = _bn2d_ws(N, Cc, HW, dev)
    y = torch.empty_like(x)
    stat = torch.empty((2, Cc), dtype=torch.float32, device=dev)
    rc = lib().rslo_bn2d_fwd_local(_ptr(x, torch.float32, "x"), _ptr(res, torch.float32, "res"), _dp(gamma), _dp(beta), N,
                                   Cc, HW, eps, momentum, slope, _dp(running_mean), _dp(running_var), stat.data_ptr(),
                                   stat.data_ptr() + 4 * Cc, y.data_ptr(), ws.data_ptr(), wsb, _stream())
    if rc:
        _chk(rc, "rslo_bn2d_fwd_local")
    return y, stat[0], stat[1]


def bn2d_bwd_local(dy, y, x, gamma, mean, invstd, slope, has_act, want_res, want_affine=True):
    N, Cc, H, W = x.shape
    dev = x.device
    HW = H * W
    ws, wsb = _bn2d_ws(N, Cc, HW, dev)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_res else None
    if want_affine:
        dgb = torch.empty((2, Cc), dtype=torch.float32, device=dev)
        dgamma, dbeta = dgb[0], dgb[1]
        pg, pb = dgb.data_ptr(), dgb.data_ptr() + 4 * Cc
    else:
        dgamma = dbeta = pg = pb = None
    rc = lib().rslo_bn2d_bwd_local(_ptr(dy, torch.float32, "dy"), _dp(y), x.data_ptr(), _dp(gamma), mean.data_ptr(),
                                   invstd.data_ptr(), N, Cc, HW, slope, 1 if has_act else 0, dx.data_ptr(), _dp(dres), pg,
                                   pb, ws.data_ptr(), wsb, _stream())
    if rc:
        _chk(rc, "rslo_bn2d_bwd_local")
    return dx, dres, dgamma, dbeta


_bn_peer_ok = {}


def bn2d_peer_supported(N, Cc, HW):
    key = (N, Cc, HW)
    v = _bn_peer_ok.get(key)
    if v is None:
        v = _bn_peer_ok[key] = bool(lib().rslo_bn2d_peer_supported(N, Cc, HW))
    return v


def bn2d_fwd_peer(comm, x, res, gamma, beta, running_mean, running_var, momentum, eps, slope):
    """Multi-rank y = act(BN_train(x) + res) in ONE launch: the per-channel sums meet the other ranks' inside the kernel
    (rslo_bn2d_fwd_peer; comm = rslo_amd.peer.PeerComm).  -> y, save_mean, save_invstd, count_all (device double [1])."""
    N, Cc, H, W = x.shape
    dev = x.device
    y = torch.empty_like(x)
    stat = torch.empty((2, Cc), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.float64, device=dev)
    rc = lib().rslo_bn2d_fwd_peer(comm.handle, _ptr(x, torch.float32, "x"), _ptr(res, torch.float32, "res"), _dp(gamma),
                                  _dp(beta), N, Cc, H * W, eps, momentum, slope, _dp(running_mean), _dp(running_var),
                                  stat.data_ptr(), stat.data_ptr() + 4 * Cc, cnt.data_ptr(), y.data_ptr(), _stream())
    if rc:
        _chk(rc, "rslo_bn2d_fwd_peer")
    return y, stat[0], stat[1], cnt


def bn2d_bwd_peer(comm, dy, y, x, gamma, mean, invstd, count_all, slope, has_act, want_res, want_affine=True):
    N, Cc, H, W = x.shape
    dev = x.device
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_res else None
    if want_affine:
        dgb = torch.empty((2, Cc), dtype=torch.float32, device=dev)
        dgamma, dbeta = dgb[0], dgb[1]
        pg, pb = dgb.data_ptr(), dgb.data_ptr() + 4 * Cc
    else:
        dgamma = dbeta = pg = pb = None
    rc = lib().rslo_bn2d_bwd_peer(comm.handle, _ptr(dy, torch.float32, "dy"), _dp(y), x.data_ptr(), _dp(gamma),
                                  mean.data_ptr(), invstd.data_ptr(), count_all.data_ptr(), N, Cc, H * W, slope,
                                  1 if has_act else 0, dx.data_ptr(), _dp(dres), pg, pb, _stream())
    if rc:
        _chk(rc, "rslo_bn2d_bwd_peer")
    return dx, dres, dgamma, dbeta


def roi_threshold(dist, counts, ratio):
    """dist [B,N] (+inf padded), counts int32 [B] or None -> thr [B] = max(kth(dist_b, 1 + int(cnt_b * ratio)), 1)."""
    B, N = dist.shape
    thr = torch.empty((B,), dtype=torch.float32, device=dist.device)
    _chk(lib().rslo_roi_threshold(_ptr(dist, torch.float32, "dist"), B, N, _ptr(counts, torch.int32, "counts"),
                                  float(ratio), _ptr(thr), _stream()), "rslo_roi_threshold")
    return thr


# --------------------------------------------------------------------------------------
# local -> global + ego-motion vote
# --------------------------------------------------------------------------------------
_vote_done = {}


def vote_fwd(tq_map, t_conf, r_conf, origin, vsize):
    B, _, H, W = tq_map.shape
    dev = tq_map.device
    done = _vote_done.get(dev)
    if done is None or done.numel() < B:
        done = _vote_done[dev] = torch.zeros((max(B, 64),), dtype=torch.int32, device=dev)
    wsb = lib().rslo_vote_ws_bytes(B, H, W)
    ws = _ws(wsb, dev)
    tq_g = torch.empty_like(tq_map)
    odom = torch.empty((B, 7), dtype=torch.float32, device=dev)
    sums = torch.empty((B, 2), dtype=torch.float32, device=dev)
    _chk(lib().rslo_vote_fwd(_ptr(tq_map, torch.float32, "tq_map"), _ptr(t_conf, torch.float32, "t_conf"),
                             _ptr(r_conf, torch.float32, "r_conf"), B, H, W, _F3(*[float(v) for v in origin]),
                             _F3(*[float(v) for v in vsize]), _ptr(ws), wsb, _ptr(done), _ptr(tq_g), _ptr(odom),
                             _ptr(sums), _stream()), "rslo_vote_fwd")
    return tq_g, odom, sums


def vote_bwd(tq_map, t_conf, r_conf, origin, vsize, odom, sums, g_odom):
    B, _, H, W = tq_map.shape
    d_tq = torch.empty_like(tq_map)
    d_tc = torch.empty_like(t_conf)
    d_rc = torch.empty_like(r_conf)
    _chk(lib().rslo_vote_bwd(_ptr(tq_map, torch.float32, "tq_map"), _ptr(t_conf, torch.float32, "t_conf"),
                             _ptr(r_conf, torch.float32, "r_conf"), B, H, W, _F3(*[float(v) for v in origin]),
                             _F3(*[float(v) for v in vsize]), _ptr(odom), _ptr(sums), _ptr(g_odom, torch.float32, "g"),
                             _ptr(d_tq), _ptr(d_tc), _ptr(d_rc), _stream()), "rslo_vote_bwd")
    return d_tq, d_tc, d_rc


# --------------------------------------------------------------------------------------
# dense 3x3 conv2d weight gradient (BEV head)
# --------------------------------------------------------------------------------------
def conv2d_wgrad_supported(cin, cout, H, W, stride):
    return bool(lib().rslo_conv2d_wgrad_supported(int(cin), int(cout), int(H), int(W), int(stride)))


_c2w_ws_bytes = {}


def conv2d_wgrad(x, dout, stride=1, want_bias=False, lp=False):
    """x [B,cin,H,W], dout [B,cout,Ho,Wo] (contiguous NCHW fp32) -> dW [cout,cin,3,3] of a 3x3 / padding-1 conv;
    want_bias (stride 1): -> (dW, dbias [cout]) with the bias gradient from the same pass.  lp: bf16 operands (C4)."""
    B, cin, H, W = x.shape
    cout = dout.shape[1]
    key = (B, cin, cout, H, W, stride)
    wsb = _c2w_ws_bytes.get(key)
    if wsb is None:
        wsb = _c2w_ws_bytes[key] = lib().rslo_conv2d_wgrad_ws_bytes(B, cin, cout, H, W, stride)
    if wsb == 0:
        raise RsloHipError("rslo_conv2d_wgrad: unsupported shape %s -> %s stride %d" % (tuple(x.shape), tuple(dout.shape), stride))
    dev = x.device
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    dW = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if want_bias else None
    fn = lib().rslo_conv2d_wgrad_bf16 if lp else lib().rslo_conv2d_wgrad
    rc = fn(_ptr(x, torch.float32, "x"), _ptr(dout, torch.float32, "dout"), B, cin, cout, H, W,
            stride, dW.data_ptr(), _dp(db), ws.data_ptr(), wsb, _stream())
    if rc:
        _chk(rc, "rslo_conv2d_wgrad")
    _keep_for_reduce(ws)
    return (dW, db) if want_bias else dW


def conv1x1s2_wgrad(x, dout):
    """x [B,cin,H,W], dout [B,cout,Ho,Wo] -> dW [cout,cin,1,1] of a 1x1 / stride-2 / padding-0 conv (rslo_conv1x1s2_wgrad);
    None when the shape is outside the kernel's range (the caller falls back to a batched GEMM)."""
    B, cin, H, W = x.shape
    cout = dout.shape[1]
    key = (B, cin, cout, H, W, "1x1s2")
    wsb = _c2w_ws_bytes.get(key)
    if wsb is None:
        wsb = _c2w_ws_bytes[key] = lib().rslo_conv1x1s2_wgrad_ws_bytes(B, cin, cout, H, W)
    if wsb == 0:
        return None
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    dW = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=x.device)
    rc = lib().rslo_conv1x1s2_wgrad(_ptr(x, torch.float32, "x"), _ptr(dout, torch.float32, "dout"), B, cin, cout, H, W,
                                    dW.data_ptr(), ws.data_ptr(), wsb, _stream())
    if rc:
        _chk(rc, "rslo_conv1x1s2_wgrad")
    _keep_for_reduce(ws)
    return dW


def conv2d_fwd_supported(cin, cout, H, W):
    return bool(lib().rslo_conv2d_fwd_supported(int(cin), int(cout), int(H), int(W)))


def conv2d_wsplit(w, transpose=False):
    """w [cout,cin,3,3] fp32 -> split-bf16 MFMA operands (int16 tensor) for conv2d_fwd; transpose=True: data gradient."""
    cout, cin = w.shape[0], w.shape[1]
    ws = torch.empty((lib().rslo_conv2d_wsplit_bytes(cin, cout) // 2,), dtype=torch.int16, device=w.device)
    _chk(lib().rslo_conv2d_wsplit(_ptr(w, torch.float32, "w"), cin, cout, 1 if transpose else 0, _ptr(ws), _stream()),
         "rslo_conv2d_wsplit")
    return ws


class Conv2dSplitDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ws_fwd", C.c_void_p), ("ws_dgrad", C.c_void_p), ("cin", C.c_int32),
                ("cout", C.c_int32), ("ntap", C.c_int32), ("reserved", C.c_int32)]


def conv2d_wsplit_many(weights):
    """weights: list of [cout,cin,3,3] (or [cout,cin,1,1]) fp32 CUDA tensors.  Returns (plan, [(ws_fwd, ws_dgrad) per
    layer]); call conv2d_wsplit_run(plan) after every weight update: it refreshes all operands in one launch."""
    dev = weights[0].device
    sizes = [lib().rslo_conv2d_wsplit_bytes(w.shape[1], w.shape[0]) // 2 * (w.shape[2] * w.shape[3]) // 9 for w in weights]
    pool = torch.empty((2 * sum(sizes),), dtype=torch.int16, device=dev)
    views, off = [], 0
    arr = (Conv2dSplitDesc * len(weights))()
    for i, (w, n) in enumerate(zip(weights, sizes)):
        f, t = pool[off:off + n], pool[off + n:off + 2 * n]
        off += 2 * n
        views.append((f, t))
        arr[i] = Conv2dSplitDesc(_ptr(w, torch.float32, "w"), f.data_ptr(), t.data_ptr(), w.shape[1], w.shape[0],
                                 w.shape[2] * w.shape[3], 0)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    table = host.to(dev)
    plan = {"table": table, "n": len(weights), "max": max(int(w.numel()) for w in weights), "pool": pool,
            "ptrs": [w.data_ptr() for w in weights]}
    return plan, views


def conv2d_wsplit_run(plan):
    _chk(lib().rslo_conv2d_wsplit_many(_ptr(plan["table"]), plan["n"], plan["max"], _stream()), "rslo_conv2d_wsplit_many")


def conv2d_s2_supported(cin, cout, ksize):
    return bool(lib().rslo_conv2d_s2_supported(int(cin), int(cout), int(ksize)))


def conv2d_wsplit_k(w, transpose=False):
    """w [cout,cin,k,k] (k = 1 or 3) fp32 -> split-bf16 operands for conv2d_fwd_s2 (transpose=True: conv2d_dgrad_s2)."""
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    ws = torch.empty((3 * k * k * cin * cout,), dtype=torch.int16, device=w.device)
    _chk(lib().rslo_conv2d_wsplit_k(_ptr(w, torch.float32, "w"), cin, cout, k, 1 if transpose else 0, _ptr(ws), _stream()),
         "rslo_conv2d_wsplit_k")
    return ws


def conv2d_fwd_s2(x, ws, cout, ksize):
    """x [B,cin,H,W] contiguous fp32 -> [B,cout,Ho,Wo]: 3x3 / padding 1 or 1x1 / padding 0, stride 2, no bias."""
    B, cin, H, W = x.shape
    out = torch.empty((B, cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    rc = lib().rslo_conv2d_fwd_s2(_ptr(x, torch.float32, "x"), ws.data_ptr(), B, cin, cout, H, W, ksize, out.data_ptr(),
                                  _stream())
    if rc:
        _chk(rc, "rslo_conv2d_fwd_s2")
    return out


def conv2d_dgrad_s2(dy, ws_t, cin, H, W, ksize, residual=None, inplace=False):
    """dy [B,cout,Ho,Wo] -> dx [B,cin,H,W] of the stride-2 layer; ws_t = the transpose=True operand.
    residual [B,cin,H,W]: added in the epilogue (rslo_conv2d_dgrad_s2_add); inplace (ksize 1): the gradient is added INTO
    residual, which is returned -- only the pixels (2y, 2x) are read and written."""
    B, cout = dy.shape[0], dy.shape[1]
    shape = (B, cin, H, W)
    if residual is not None and (tuple(residual.shape) != shape or not residual.is_contiguous()):
        raise ValueError("conv2d_dgrad_s2: residual must be a contiguous %s tensor" % (shape,))
    if inplace and (residual is None or ksize != 1):
        raise ValueError("conv2d_dgrad_s2: inplace needs a residual and ksize 1")
    dx = residual if inplace else torch.empty(shape, dtype=torch.float32, device=dy.device)
    rc = lib().rslo_conv2d_dgrad_s2_add(_ptr(dy, torch.float32, "dy"), ws_t.data_ptr(),
                                        _ptr(residual, torch.float32, "residual") if residual is not None else None,
                                        B, cin, cout, H, W, ksize, dx.data_ptr(), _stream())
    if rc:
        _chk(rc, "rslo_conv2d_dgrad_s2")
    return dx


def conv2d_fwd(x, ws, bias, cout, lp=False, residual=None):
    """x [B,cin,H,W] contiguous fp32, ws from conv2d_wsplit -> [B,cout,H,W] (3x3, stride 1, padding 1).
    lp: bf16 operands, fp32 accumulation (C4).  residual [B,cout,H,W]: added in the epilogue (rslo_conv2d_fwd_add)."""
    B, cin, H, W = x.shape
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    if residual is not None:
        if tuple(residual.shape) != tuple(out.shape) or not residual.is_contiguous():
            raise ValueError("conv2d_fwd: residual must be a contiguous %s tensor" % (tuple(out.shape),))
        fn = lib().rslo_conv2d_fwd_add_bf16 if lp else lib().rslo_conv2d_fwd_add
        rc = fn(_ptr(x, torch.float32, "x"), ws.data_ptr(), _dp(bias), _ptr(residual, torch.float32, "residual"), B, cin,
                cout, H, W, out.data_ptr(), _stream())
        if rc:
            _chk(rc, "rslo_conv2d_fwd_add")
        return out
    fn = lib().rslo_conv2d_fwd_bf16 if lp else lib().rslo_conv2d_fwd
    rc = fn(_ptr(x, torch.float32, "x"), ws.data_ptr(), _dp(bias), B, cin, cout, H, W, out.data_ptr(),
            _stream())
    if rc:
        _chk(rc, "rslo_conv2d_fwd")
    return out


def opl_from_nchw(x):
    """x [B,C,H,W] contiguous fp32 (C % 8 == 0) -> operand planes [B, C/8, 3, H, W, 8] bf16 bit patterns (int16):
    the exact hi / mid / lo split, one 16-byte piece per (pixel, channel octet, plane) -- what conv2d_fwd_p stages."""
    B, Cc, H, W = x.shape
    pl = torch.empty((B, Cc // 8, 3, H, W, 8), dtype=torch.int16, device=x.device)
    _chk(lib().rslo_opl_from_nchw(_ptr(x, torch.float32, "x"), B, Cc, H, W, pl.data_ptr(), _stream()), "rslo_opl_from_nchw")
    return pl


def conv2d_fwd_p_supported(cin, cout, H, W):
    return bool(lib().rslo_conv2d_fwd_p_supported(int(cin), int(cout), int(H), int(W)))


def conv2d_fwd_p(planes, ws, bias, cout, residual=None):
    """planes [B,cin/8,3,H,W,8] (opl_from_nchw or a BatchNorm apply kernel), ws from conv2d_wsplit -> [B,cout,H,W] fp32
    (3x3, stride 1, padding 1): the same bits as conv2d_fwd on the un-split tensor."""
    B, no, _, H, W, _ = planes.shape
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=planes.device)
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or not residual.is_contiguous()):
        raise ValueError("conv2d_fwd_p: residual must be a contiguous %s tensor" % (tuple(out.shape),))
    if not planes.is_contiguous() or planes.dtype != torch.int16:
        raise ValueError("conv2d_fwd_p: planes must be a contiguous int16 tensor")
    rc = lib().rslo_conv2d_fwd_p(planes.data_ptr(), ws.data_ptr(), _dp(bias),
                                 _ptr(residual, torch.float32, "residual") if residual is not None else None,
                                 B, no * 8, cout, H, W, out.data_ptr(), _stream())
    if rc:
        _chk(rc, "rslo_conv2d_fwd_p")
    return out


def conv1x1_supported(cin, cout):
    return bool(lib().rslo_conv1x1_supported(int(cin), int(cout)))


def conv1x1_fwd(x, w, bias):
    """x [B,cin,H,W], w [cout,cin,1,1] (cout <= 8), bias [cout] or None -> [B,cout,H,W]."""
    B, cin, H, W = x.shape
    cout = w.shape[0]
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    rc = lib().rslo_conv1x1_fwd(_ptr(x, torch.float32, "x"), _ptr(w, torch.float32, "w"), _dp(bias), B, cin, cout, H * W,
                                out.data_ptr(), _stream())
    if rc:
        _chk(rc, "rslo_conv1x1_fwd")
    return out


def conv1x1_dgrad(dy, w):
    B, cout, H, W = dy.shape
    cin = w.shape[1]
    dx = torch.empty((B, cin, H, W), dtype=torch.float32, device=dy.device)
    rc = lib().rslo_conv1x1_dgrad(_ptr(dy, torch.float32, "dy"), _ptr(w, torch.float32, "w"), B, cin, cout, H * W,
                                  dx.data_ptr(), _stream())
    if rc:
        _chk(rc, "rslo_conv1x1_dgrad")
    return dx


def conv1x1_wgrad(x, dy, want_bias=True):
    B, cin, H, W = x.shape
    cout = dy.shape[1]
    dev = x.device
    wsb = lib().rslo_conv1x1_wgrad_ws_bytes(B, cin, cout, H * W)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    dW = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if want_bias else None
    rc = lib().rslo_conv1x1_wgrad(_ptr(x, torch.float32, "x"), _ptr(dy, torch.float32, "dy"), B, cin, cout, H * W,
                                  dW.data_ptr(), _dp(db), ws.data_ptr(), wsb, _stream())
    if rc:
        _chk(rc, "rslo_conv1x1_wgrad")
    return dW, db


# --------------------------------------------------------------------------------------
# per-pair pose algebra
# --------------------------------------------------------------------------------------
def quat_to_rot(q):
    B = q.shape[0]
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=q.device)
    _chk(lib().rslo_quat_to_rot(_ptr(q, torch.float32, "q"), B, _ptr(R), _stream()), "rslo_quat_to_rot")
    return R


def quat_to_rot_bwd(q, gR):
    B = q.shape[0]
    gq = torch.empty((B, 4), dtype=torch.float32, device=q.device)
    _chk(lib().rslo_quat_to_rot_bwd(_ptr(q, torch.float32, "q"), _ptr(gR, torch.float32, "gR"), B, _ptr(gq), _stream()),
         "rslo_quat_to_rot_bwd")
    return gq


def pose_targets(res_r, res_t, R_pred, T_pred, with_tq=False):
    """-> (q* [B,4] wxyz, t* [B,3]); with_tq: also the rows (t*, q*) [B,7] from the same launch."""
    B = res_r.shape[0]
    rot = torch.empty((B, 4), dtype=torch.float32, device=res_r.device)
    trans = torch.empty((B, 3), dtype=torch.float32, device=res_r.device)
    tq = torch.empty((B, 7), dtype=torch.float32, device=res_r.device) if with_tq else None
    _chk(lib().rslo_pose_targets_tq(_ptr(res_r, torch.float32, "res_r"), _ptr(res_t, torch.float32, "res_t"),
                                    _ptr(R_pred, torch.float32, "R_pred"), _ptr(T_pred, torch.float32, "T_pred"), B,
                                    _ptr(rot), _ptr(trans), _ptr(tq), _stream()), "rslo_pose_targets_tq")
    return (rot, trans, tq) if with_tq else (rot, trans)
