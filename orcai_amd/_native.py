"""ctypes binding of liborcai_hip.so (include/orcai_hip.h).

There is NO CPU fallback: if the library is missing or a call fails the error is raised.
Device memory, streams and process groups come from PyTorch-ROCm; the library itself only
sees raw device pointers and a hipStream_t.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "liborcai_hip.so"

_lib = None

c_i64 = C.c_int64
c_f32p = C.c_void_p  # device pointers travel as void*

class UnpackDesc(C.Structure):
    """orcai_unpack_desc of include/orcai_hip.h."""

    _fields_ = [("src", C.c_void_p), ("ld_src", C.c_int), ("col_off", C.c_int), ("rows", C.c_int), ("G", C.c_void_p), ("W", C.c_void_p), ("l2g", C.c_float)]


_SIGNATURES = {
    "orcai_version": (C.c_char_p, []),
    "orcai_frontend_workspace_bytes": (C.c_size_t, []),
    "orcai_frontend_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orcai_stft_db": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_int, c_i64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_stft_blocks": (C.c_int, [C.c_int]),
    "orcai_stft_occupancy": (C.c_int, []),
    "orcai_hist_level1": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_quantile_select": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_frontend_finalize": (C.c_int, [C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_clip_normalize": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_db_reference": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_crop_transpose": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_frontend_stats_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]),
    "orcai_conv0_bn_relu": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_sepconv_bn": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_pool_res_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "orcai_padded_width": (C.c_int, [C.c_int, C.c_int]),
    "orcai_sepconv_tile_mode": (C.c_int, [C.c_int]),
    "orcai_entry_windows": (C.c_int, [C.c_int]),
    "orcai_entry_tile": (C.c_int, [C.c_int]),
    "orcai_conv0_sepconv": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_gemm_bias_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "orcai_lstm_recurrent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_dense_sigmoid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_overlap_average": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_gemm_strided": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "orcai_colsum": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "orcai_bn_rows_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_bn_rows_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_bn_rows_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_dropout_mask": (C.c_int, [C.c_void_p, c_i64, C.c_uint64, C.c_float, C.c_void_p]),
    "orcai_mask_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_relu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_masked_bce": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_masked_bce_w": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "orcai_l2_value": (C.c_int, [C.c_void_p, c_i64, C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "orcai_lstm_split": (C.c_int, [C.c_int]),
    "orcai_lstm_train_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_h_lstm_train_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_lstm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_h_lstm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_lstm_hprev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_conv0_affine": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_gather_snippets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_downsample_labels": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "orcai_freq_mean_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_conv1d_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p] * 3),
    "orcai_bn_bwd_pointwise": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3),
    "orcai_pw_wgrad_tiles": (C.c_int, [C.c_int]),
    "orcai_bn_bwd_pointwise_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [c_i64, C.c_void_p]),
    "orcai_h_bn_bwd_pointwise_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [c_i64, C.c_void_p]),
    "orcai_pool_res_add_bn": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p]),
    "orcai_pool_bwd_bn": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_pool_bwd_bn_bias": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 4),
    "orcai_h_pool_bwd_bn_bias": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 4),
    "orcai_conv0_bn_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 5),
    "orcai_conv0_stats": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_int] * 4 + [C.c_void_p] * 5),
    "orcai_conv0_affine_bn": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_conv0_bn_bwd_x": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 5 + [c_i64, C.c_void_p]),
    "orcai_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_freq_mean": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_conv1d_sigmoid": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "orcai_sepconv_planes": (C.c_int, [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "orcai_sepconv_planes_u": (C.c_int, [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_sepconv_planes_stats": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4),
    "orcai_dw_wgrad_march": (C.c_int, [C.c_int]),
    "orcai_pool_vertical": (C.c_int, [C.c_int]),
    "orcai_pool_fused": (C.c_int, [C.c_int]),
    "orcai_sepconv_pool_res": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 8 + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]),
    "orcai_dw_bwd_fused": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_dw_bwd_fused_conv0": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] * 9 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_conv0_bn_bwd_x_ready": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 5 + [c_i64, C.c_void_p]),
    "orcai_conv0_stats_march": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_int] * 3 + [C.c_void_p] * 5),
    "orcai_conv0_march": (C.c_int, [C.c_int]),
    "orcai_h_dw_bwd_fused": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_h_dw_bwd_fused_res": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_h_conv0_bn_bwd_ready": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 5),
    "orcai_dw_wgrad_bn": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_sepconv_planes_stats_bn": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4),
    "orcai_sepconv_planes_epi": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_bn_finish_sharded": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 3),
    "orcai_h_sepconv_stats": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4),
    "orcai_h_sepconv_stats_bn": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4),
    "orcai_h_bn_finish_sharded": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 3),
    "orcai_bn_planes_stats": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4),
    "orcai_bn_planes_apply": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_bn_planes_bwd": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_int] + [C.c_void_p] * 5),
    "orcai_planes_sum": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "orcai_pool_bwd": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "orcai_outer_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "orcai_outer_reduce_pixels": (C.c_int, [C.c_int]),
    "orcai_dw_wgrad": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]),
    "orcai_conv0_wgrad": (C.c_int, [C.c_void_p, c_i64, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "orcai_feat_to_planes": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "orcai_planes_relu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_resample_polyphase": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, c_i64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "orcai_h_conv0_affine": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_h_conv0_affine_bn": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_h_sepconv": (C.c_int, [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_h_pool_res_add": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p]),
    "orcai_h_gemm_bias_act": (C.c_int, [C.c_void_p] * 6 + [c_i64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "orcai_h_bn_planes_stats": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4),
    "orcai_h_planes_sum": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "orcai_h_bn_planes_apply": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_h_bn_bwd_pointwise": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3),
    "orcai_h_pool_bwd_bn": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_h_outer_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "orcai_h_dw_wgrad": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]),
    "orcai_h_conv0_bn_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 5),
    "orcai_h_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orcai_h_feat_to_planes": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "orcai_h_planes_relu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_void_p]),
    "orcai_dropout_mask_dev": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]),
    "orcai_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    "orcai_counter_advance": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orcai_step_ok": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_scratch_arena": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "orcai_arena_take": (C.c_int, [C.c_void_p, C.c_size_t]),
    "orcai_poison_if_nonfinite": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "orcai_adam_step_guarded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_ema_update_guarded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_counter_advance_guarded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_pack_lstm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orcai_unpack_lstm_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "orcai_profile_bracket": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orcai_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "orcai_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "orcai_event_destroy": (C.c_int, [C.c_void_p]),
    "orcai_unpack_lstm_grads": (C.c_int, [C.POINTER(UnpackDesc), C.c_int, C.c_int, C.c_void_p]),
    "orcai_l2_values": (C.c_int, [C.c_void_p, C.POINTER(c_i64), C.POINTER(c_i64), C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "orcai_ema_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "orcai_make_spectrogram": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_int, c_i64, C.c_int, c_i64, c_i64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
}


class NativeLibraryError(RuntimeError):
    pass


def exported_symbols() -> list[str]:
    """Every symbol include/orcai_hip.h declares (kept in sync with the header by tests/test_capi_symbols.py)."""
    return list(_SIGNATURES)


def lib() -> C.CDLL:
    """Load liborcai_hip.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64; it must be in the process BEFORE this library is dlopen'ed so both
        # share ONE HIP runtime (otherwise /opt/rocm's copy is pulled in and torch's device pointers are
        # foreign to it: hipErrorNoDevice).
        import torch  # noqa: F401
        if not LIB_PATH.exists():
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python -m orcai_amd.build` (needs hipcc). " "orcai_amd has no CPU fallback."
            )
        try:
            handle = C.CDLL(str(LIB_PATH))
        except OSError as e:  # missing ROCm runtime etc.
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


E_BADARG = -1  # ORCAI_E_BADARG (include/orcai_hip.h)
E_UNSUPPORTED = -2  # ORCAI_E_UNSUPPORTED


def check(code: int, what: str) -> None:
    if code == 0:
        return
    if code == -1:
        raise ValueError(f"{what}: bad argument (ORCAI_E_BADARG)")
    if code == -2:
        raise NotImplementedError(f"{what}: unsupported parameter combination on the HIP path (ORCAI_E_UNSUPPORTED)")
    raise RuntimeError(f"{what}: HIP error {code}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (must be contiguous)."""
    assert t.is_contiguous(), "native ops need contiguous tensors"
    return t.data_ptr()


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
