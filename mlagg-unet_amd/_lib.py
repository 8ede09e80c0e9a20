"""ctypes binding of libmlagg_hip.so (C ABI: include/mlagg_hip.h).

There is no fallback: if the shared library is absent or a symbol is missing the import of any op
raises, and every non-zero return code of an entry point becomes RuntimeError (the exception type
nnU-Net's trainers handle, reference nnUNetTrainerBenchmark_5epochs.py:25-29)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libmlagg_hip.so")

_F = ctypes.c_void_p      # device pointer
_I = ctypes.c_int
_S = ctypes.c_void_p      # hipStream_t
_SZ = ctypes.c_size_t
_FL = ctypes.c_float

# name -> (restype, argtypes); mirrors include/mlagg_hip.h one to one
SIGNATURES = {
    "mlagg_version": (ctypes.c_char_p, []),
    "mlagg_error_string": (ctypes.c_char_p, [_I]),
    "mlagg_profile_kernel_count": (_I, []),
    "mlagg_profile_kernel_name": (ctypes.c_char_p, [_I]),
    "mlagg_profile_select": (_I, [_I]),
    "mlagg_profile_collect": (_I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "mlagg_selscan_state_floats": (_SZ, [_I, _I, _I, _I]),
    "mlagg_selscan_fwd": (_I, [_F] * 9 + [_I] * 6 + [_S]),
    "mlagg_selscan_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I]),
    "mlagg_selscan_bwd": (_I, [_F] * 17 + [_I] * 6 + [_S]),
    "mlagg_selscan_lowrank_fwd": (_I, [_F] * 3 + [_I] + [_F] * 7 + [_I] * 6 + [_S]),
    "mlagg_selscan_lowrank_bwd": (_I, [_F] * 3 + [_I] + [_F] * 16 + [_I] * 6 + [_S]),
    "mlagg_msmm_scan_supported": (_I, [_I] * 5),
    "mlagg_msmm_scan_state_floats": (_SZ, [_I, _I]),
    "mlagg_msmm_scan_fwd_workspace_floats": (_SZ, [_I, _I]),
    "mlagg_msmm_scan_bwd_workspace_floats": (_SZ, [_I, _I]),
    "mlagg_msmm_scan_fwd": (_I, [_F] * 10 + [_I, _I, _S]),
    "mlagg_msmm_scan_bwd": (_I, [_F] * 16 + [_I, _I, _S]),
    "mlagg_local_attn_fwd": (_I, [_F, _I, _F, _I, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _FL, _S]),
    "mlagg_local_attn_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I]),
    "mlagg_local_attn_bwd": (_I, [_F, _I, _F, _I, _F, _F, _F, _F, _I, _F, _I, _F, _I, _F, _F, _F, _F, _F,
                                  _I, _I, _I, _I, _FL, _S]),
    "mlagg_pooled_attn_fwd": (_I, [_F, _I, _F, _I, _F, _I, _F, _F, _F, _I, _F, _F, _I, _I, _I, _I, _FL, _S]),
    "mlagg_pooled_attn_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I]),
    "mlagg_pooled_attn_bwd": (_I, [_F, _I, _F, _I, _F, _I, _F, _F, _F, _I, _F, _F, _F, _I, _F, _I, _F, _I, _F, _F,
                                   _F, _I, _I, _I, _I, _FL, _S]),
    "mlagg_pooled_attn_lp_fwd": (_I, [_F, _I, _F, _I, _F, _I, _F, _F, _F, _I, _F, _F, _F, _I, _I, _I, _I, _FL, _I, _S]),
    "mlagg_pooled_attn_lp_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I]),
    "mlagg_pooled_attn_lp_bwd": (_I, [_F, _I, _F, _I, _F, _I, _F, _F, _F, _I, _F, _F, _F, _F, _I, _F, _I, _F, _I, _F, _F, _F,
                                      _I, _I, _I, _I, _FL, _I, _S]),
    "mlagg_dwconv3x3_fwd": (_I, [_F, _I, _F, _F, _F, _F, _I, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_dwconv3x3_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I]),
    "mlagg_dwconv3x3_bwd": (_I, [_F, _I, _F, _F, _I, _F, _F, _I, _F, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_dwconv3x3_gated_fwd": (_I, [_F, _I, _F, _F, _F, _I, _F, _I, _F, _I, _I, _I, _I, _S]),
    "mlagg_dwconv3x3_gated_bwd": (_I, [_F, _I, _F, _F, _I, _F, _F, _I, _F, _I, _F, _I, _F, _F, _F, _I, _I, _I, _I, _S]),
    "mlagg_dwconv3d_fwd": (_I, [_F, _I, _F, _F, _F, _I, _F, _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_dwconv3d_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I, _I]),
    "mlagg_dwconv3d_bwd": (_I, [_F, _I, _F, _F, _I, _F, _F, _I, _F, _F, _F, _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_linear_wgrad_workspace_floats": (_SZ, [_I, _I, _I]),
    "mlagg_linear_wgrad": (_I, [_F, _I, _F, _I, _F, _F, _F, _I, _I, _I, _S]),
    "mlagg_linear_wgrad_x3": (_I, [_F, _I, _F, _I, _F, _F, _F, _I, _I, _I, _S]),
    "mlagg_layernorm_supported": (_I, [_I]),
    "mlagg_layernorm_fwd": (_I, [_F, _I, _F, _F, _F, _F, _I, _I, _FL, _S]),
    "mlagg_layernorm_bwd_workspace_floats": (_SZ, [_I, _I]),
    "mlagg_layernorm_bwd": (_I, [_F, _I, _F, _I, _F, _F, _F, _F, _F, _F, _I, _I, _S]),
    "mlagg_residual_layernorm_fwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, ctypes.c_float, _S]),
    "mlagg_residual_layernorm_bwd": (_I, [_F, _F, _I, _F, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _S]),
    "mlagg_dwconv3x3_nchw_fwd": (_I, [_F, _F, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_dwconv3x3_nchw_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I, _I]),
    "mlagg_dwconv3x3_nchw_bwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_dwconv3x3_nchw_bwd_res": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_cross_scan": (_I, [_F, _I, _I, _F, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _I, _S]),
    "mlagg_cross_merge": (_I, [_F, _F, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _I, _S]),
    "mlagg_diff_lambda_fwd": (_I, [_F, _F, _F, _F, ctypes.c_float, _I, _F, _F, _S]),
    "mlagg_diff_lambda_bwd": (_I, [_F, _F, _F, _F, _F, _F, _I, _F, _F, _F, _F, _S]),
    "mlagg_scaled_residual": (_I, [_F, _F, _F, _F, _I, ctypes.c_long, _S]),
    "mlagg_dice_ce_max_classes": (_I, []),
    "mlagg_dice_ce_stats_workspace_floats": (_SZ, [_I, _I, ctypes.c_long]),
    "mlagg_dice_ce_stats": (_I, [_F, _F, _F, _F, _F, _F, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_dice_ce_grad": (_I, [_F, _F, _F, _F, _F, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_channel_sum_workspace_floats": (_SZ, [_I, _I]),
    "mlagg_channel_sum": (_I, [_F, _F, _F, _I, _I, ctypes.c_long, _S]),
    "mlagg_column_sum_workspace_floats": (ctypes.c_size_t, [_I, _I]),
    "mlagg_column_sum": (_I, [_F, _I, _F, _F, _I, _I, _S]),
    "mlagg_plane_norm_fwd_workspace_floats": (_SZ, [_I, _I, ctypes.c_long]),
    "mlagg_plane_norm_fwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _I, _I, ctypes.c_long, ctypes.c_float, _I, ctypes.c_float, _I, _I, _I, _S]),
    "mlagg_plane_norm_bwd_workspace_floats": (_SZ, [_I, _I, ctypes.c_long]),
    "mlagg_plane_norm_bwd": (_I, [_F] * 11 + [_I, _I, ctypes.c_long, _I, ctypes.c_float, _I, _I, _I, _S]),
    "mlagg_plane_norm_bwd_strided": (_I, [_F, _F, ctypes.c_long] + [_F] * 9 + [_I, _I, ctypes.c_long, _I, ctypes.c_float, _I, _I, _I, _S]),
    "mlagg_adamw_chunk_elements": (_I, []),
    "mlagg_adamw_clip_step": (_I, [_F, _F, _I, _F] + [ctypes.c_float] * 6 + [_I, _S]),
    "mlagg_adamw_clip_step_dev": (_I, [_F, _F, _I, _F, _F, _F] + [ctypes.c_float] * 5 + [_S]),
    "mlagg_transpose_2d": (_I, [_F, ctypes.c_long, _F, _I, _I, _I, _S]),
    "mlagg_transpose_2d_into": (_I, [_F, ctypes.c_long, _F, ctypes.c_long, _I, _I, _I, _S]),
    "mlagg_conv3x3_supported": (_I, [_I, _I, _I, _I]),
    "mlagg_conv3x3_workspace_bytes": (_SZ, [_I, _I]),
    "mlagg_conv3x3_fwd": (_I, [_F, ctypes.c_long, _F, _I, _F, _F, ctypes.c_long, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_conv3x3x3_supported": (_I, [_I, _I, _I, _I, _I]),
    "mlagg_conv3x3x3_workspace_bytes": (_SZ, [_I, _I]),
    "mlagg_conv3x3x3_fwd": (_I, [_F, ctypes.c_long, _F, _I, _F, _F, ctypes.c_long, _F, _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_conv3x3x3_wgrad_supported": (_I, [_I, _I, _I, _I, _I]),
    "mlagg_conv3x3x3_wgrad_workspace_floats": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "mlagg_conv3x3x3_wgrad": (_I, [_F, ctypes.c_long, _F, ctypes.c_long, _F, _F, _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_conv3x3_wgrad_supported": (_I, [_I, _I, _I, _I]),
    "mlagg_conv3x3_wgrad_workspace_floats": (_SZ, [_I, _I, _I, _I, _I]),
    "mlagg_conv3x3_wgrad": (_I, [_F, ctypes.c_long, _F, ctypes.c_long, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_conv3x3_fwd_lp": (_I, [_F, ctypes.c_long, _F, _I, _F, _F, ctypes.c_long, _F, _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_conv3x3_wgrad_lp": (_I, [_F, ctypes.c_long, _F, ctypes.c_long, _F, _F, _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_pixel_unshuffle2_strided": (_I, [_F, ctypes.c_long, _F, _I, _I, _I, _I, _S]),
    "mlagg_pixel_shuffle2": (_I, [_F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_conv1x1_fwd_acc": (_I, [_F, ctypes.c_long, _F, _F, _F, ctypes.c_long, _I, _I, _I, _I, ctypes.c_long, _I, _I, _S]),
    "mlagg_conv1x1_fwd_ragged": (_I, [_F, ctypes.c_long, _F, _F, _F, ctypes.c_long, _I, _I, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_conv1x1_fwd_lp": (_I, [_F, ctypes.c_long, _F, _F, _F, ctypes.c_long, _I, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_conv1x1_wgrad_lp": (_I, [_F, ctypes.c_long, _F, ctypes.c_long, _F, _F, _I, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_conv1x1_supported": (_I, [_I, _I, ctypes.c_long]),
    "mlagg_conv1x1_fwd": (_I, [_F, ctypes.c_long, _F, _F, _F, ctypes.c_long, _I, _I, _I, ctypes.c_long, _S]),
    "mlagg_conv1x1_wgrad_workspace_floats": (_SZ, [_I, _I, _I, ctypes.c_long]),
    "mlagg_conv1x1_wgrad": (_I, [_F, ctypes.c_long, _F, ctypes.c_long, _F, _F, _I, _I, _I, ctypes.c_long, _S]),
    "mlagg_gelu_pool_fwd": (_I, [_F, _I, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_gelu_pool_bwd": (_I, [_F, _I, _F, _F, _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_gate_fwd": (_I, [_F, _F, _F, _I, _F, ctypes.c_long, _I, _S]),
    "mlagg_gate_bwd": (_I, [_F, _I, _F, _F, _F, _I, _F, _F, _F, _I, ctypes.c_long, _I, _S]),
    "mlagg_linear_fwd": (_I, [_F, _I, _F, _F, _F, _I, _I, _I, _I, _S]),
    "mlagg_linear_dgrad": (_I, [_F, _I, _F, _F, _I, _I, _I, _I, _S]),
    "mlagg_linear_lp_fwd": (_I, [_F, _I, _F, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_linear_lp_dgrad": (_I, [_F, _I, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_weight_image_bytes": (_SZ, [_I, _I]),
    "mlagg_weight_image": (_I, [_F, _I, _F, _F, _I, _I, _S]),
    "mlagg_weight_images": (_I, [_F, _I, _I, _S]),
    "mlagg_linear_x3_supported": (_I, [_I, _I, _I]),
    "mlagg_linear_x3": (_I, [_F, _I, _F, _F, _F, _I, _F, _F, _I, _I, _I, _I, _I, _S]),
    "mlagg_flash_attn_fwd": (_I, [_F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _FL, _I, _S]),
    "mlagg_flash_attn_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I, _I]),
    "mlagg_flash_attn_bwd": (_I, [_F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _FL, _I, _S]),
    "mlagg_channel_epilogue_fwd": (_I, [_F, _F, _F, _F, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_channel_gelu_bwd": (_I, [_F, _F, _F, _F, _F, _I, _I, ctypes.c_long, _S]),
    "mlagg_channel_epilogue_lp_fwd": (_I, [_F, _I, _F, _F, _I, _F, _I, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_channel_epilogue_lp_bwd": (_I, [_F, _I, _F, _F, _I, _F, _I, _F, _I, _F, _F, _I, _I, ctypes.c_long, _I, _S]),
    "mlagg_index_scan": (_I, [_F, ctypes.c_long, _I, _F, _F, _I, _I, _I, _I, _S]),
    "mlagg_index_merge": (_I, [_F, _F, _F, ctypes.c_long, _I, _I, _I, _I, _I, _S]),
    "mlagg_block_sum": (_I, [_F, _F, ctypes.c_long, _I, _I, _S]),
    "mlagg_conv_pad_geometry": (_I, [_I, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I),
                                     ctypes.POINTER(ctypes.c_long)]),
    "mlagg_volume_pad": (_I, [_F, _F] + [_I] * 11 + [_S]),
    "mlagg_conv_taps": (_I, [_F, ctypes.c_long, ctypes.c_long, _F, ctypes.c_long, ctypes.c_long, _I, ctypes.POINTER(ctypes.c_long), _I, _F,
                             _I, _I, _I, _I, _I, _I, _S]),
    "mlagg_conv_wgrad_taps_workspace_floats": (_SZ, [_I, ctypes.c_long, _I, _I, _I]),
    "mlagg_conv_wgrad_taps": (_I, [_F, ctypes.c_long, ctypes.c_long, _F, ctypes.c_long, ctypes.c_long, ctypes.POINTER(ctypes.c_long),
                                   _I, ctypes.c_long, _I, _I, _I, _F, _I, _F, _S]),
    "mlagg_selscan1_chunk": (_I, [_I, _I, _I]),
    "mlagg_selscan1_state_floats": (_SZ, [_I, _I, _I, _I]),
    "mlagg_selscan1_fwd": (_I, [_F, ctypes.c_long] + [_F] * 5 + [_I] + [_F] * 5 + [_I] * 4 + [_S]),
    "mlagg_selscan1_bwd_workspace_floats": (_SZ, [_I, _I, _I, _I, _I]),
    "mlagg_selscan1_bwd": (_I, [_F, ctypes.c_long] + [_F] * 5 + [_I] + [_F] * 4 + [ctypes.c_long] + [_F] * 7 + [_I] * 4 + [_S]),
}

_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 into csrc/libmlagg_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the MI355X ops have no CPU or eager fallback)")
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().mlagg_error_string(int(code)).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
