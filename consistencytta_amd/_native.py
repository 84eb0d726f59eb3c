"""ctypes binding of libctta_hip.so (the C ABI declared in include/ctta.h).

PyTorch is used above this boundary only as the owner of device memory and streams: every
call passes raw `data_ptr()`s plus the current HIP stream.  There is NO fallback: if the
library is missing or a call fails, a RuntimeError is raised (the reference raises Python
exceptions at the same places -- shape asserts, load_state_dict key errors).
"""
import ctypes
from ctypes import byref  # noqa: F401  (re-exported for callers of out-parameter entry points)
import os
import subprocess
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int16, c_int32,
                    c_int64, c_size_t, c_uint8, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libctta_hip.so")
CSRC = os.path.join(_HERE, "csrc")

MAX_LEVELS = 4
MAX_UPS = 8


class Tensor(Structure):
    _fields_ = [("name", c_char_p), ("data", c_void_p), ("ndim", c_int), ("shape", c_int64 * 4)]


class UNetConfig(Structure):
    _fields_ = [
        ("in_channels", c_int), ("out_channels", c_int), ("n_levels", c_int),
        ("block_out_channels", c_int * MAX_LEVELS), ("heads", c_int * MAX_LEVELS),
        ("layers_per_block", c_int * MAX_LEVELS), ("down_cross", c_int * MAX_LEVELS),
        ("up_cross", c_int * MAX_LEVELS), ("cross_attention_dim", c_int),
        ("norm_num_groups", c_int), ("norm_eps", c_float), ("flip_sin_to_cos", c_int),
        ("freq_shift", c_float), ("guided", c_int), ("max_batch", c_int), ("height", c_int),
        ("width", c_int), ("max_text_len", c_int), ("debug_taps", c_int), ("enable_training", c_int),
    ]


class VAEConfig(Structure):
    _fields_ = [
        ("z_channels", c_int), ("embed_dim", c_int), ("ch", c_int), ("out_ch", c_int),
        ("n_levels", c_int), ("num_res_blocks", c_int), ("ch_mult", c_int * MAX_LEVELS),
        ("scale_factor", c_float), ("max_batch", c_int), ("latent_h", c_int),
        ("latent_w", c_int), ("debug_taps", c_int), ("enable_grad", c_int),
    ]


class T5Config(Structure):
    _fields_ = [
        ("vocab_size", c_int), ("d_model", c_int), ("d_kv", c_int), ("d_ff", c_int), ("num_layers", c_int),
        ("num_heads", c_int), ("rel_buckets", c_int), ("rel_max_distance", c_int), ("eps", c_float),
        ("max_batch", c_int), ("max_len", c_int),
    ]


class HifiganConfig(Structure):
    _fields_ = [
        ("num_mels", c_int), ("upsample_initial_channel", c_int), ("n_ups", c_int),
        ("n_kernels", c_int), ("upsample_rates", c_int * MAX_UPS),
        ("upsample_kernel_sizes", c_int * MAX_UPS), ("resblock_kernel_sizes", c_int * 4),
        ("resblock_dilations", (c_int * 3) * 4), ("max_batch", c_int), ("max_frames", c_int),
        ("debug_taps", c_int), ("enable_grad", c_int),
    ]


class FfnDesc(Structure):
    """ctta_ffn_desc (include/ctta.h)"""
    _fields_ = [("x", c_void_p), ("ld_x", c_int), ("M", c_int64), ("cp", c_int), ("ffp", c_int),
                ("packed", c_void_p), ("b1", c_void_p), ("b2", c_void_p), ("res", c_void_p), ("res_ld", c_int),
                ("out", c_void_p), ("ldc", c_int), ("n_valid", c_int),
                ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_d", c_int), ("ln_eps", c_float),
                ("proj_packed", c_void_p), ("proj_bias", c_void_p), ("proj_res", c_void_p), ("proj_res_ld", c_int),
                ("front_packed", c_void_p), ("front_bias", c_void_p), ("att", c_void_p), ("att_ld", c_int), ("front_k", c_int),
                ("front_res", c_void_p), ("front_res_ld", c_int), ("s2_out", c_void_p), ("s2_ld", c_int)]


class ConvDesc(Structure):
    _fields_ = [
        ("x0", c_void_p), ("c0", c_int), ("x1", c_void_p), ("c1", c_int),
        ("batch", c_int), ("hi", c_int), ("wi", c_int), ("upsample", c_int),
        ("ho", c_int), ("wo", c_int),
        ("kh", c_int), ("kw", c_int), ("stride_h", c_int), ("stride_w", c_int),
        ("pad_h", c_int), ("pad_w", c_int), ("dil_h", c_int), ("dil_w", c_int),
        ("w", c_void_p), ("k_pad", c_int), ("n", c_int),
        ("bias", c_void_p), ("bias_m", c_void_p), ("rowvec", c_void_p), ("rowvec_ld", c_int),
        ("res", c_void_p), ("res_ld", c_int),
        ("in_act", c_int), ("in_slope", c_float), ("out_act", c_int), ("out_slope", c_float),
        ("alpha", c_float), ("accumulate", c_int), ("out", c_void_p), ("ldc", c_int), ("out_f32", c_int),
        ("out2", c_void_p), ("out2_slope", c_float),
        ("out_batch_stride", c_int64), ("out_offset", c_int64), ("out_limit", c_int64),
        ("groups", c_int), ("x_group_stride", c_int64), ("w_group_stride", c_int64),
        ("out_group_stride", c_int64), ("tile", c_int), ("x_stride", c_int),
        ("gn_part", c_void_p), ("gn_groups", c_int), ("gn_hw", c_int), ("gn_part_floats", c_int64),
    ]


# name -> (restype, argtypes); every symbol declared in include/ctta.h
SIGNATURES = {
    "ctta_last_error": (c_char_p, []),
    "ctta_version": (c_int, []),
    "ctta_unet_create": (c_int, [POINTER(UNetConfig), POINTER(Tensor), c_int, c_void_p, POINTER(c_void_p)]),
    "ctta_unet_destroy": (None, [c_void_p]),
    "ctta_unet_load_weights": (c_int, [c_void_p, POINTER(Tensor), c_int, c_void_p]),
    "ctta_unet_reuse_text": (c_int, [c_void_p, c_int]),
    "ctta_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_attention_lse": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ctta_attention_bwd_inplace": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                           c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                           c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_void_p]),
    "ctta_attention_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                   c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_float, c_void_p, c_int64, c_void_p]),
    "ctta_wgrad_implicit_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ctta_wgrad_implicit_inplace": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "ctta_wgrad_tn_direct": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_wgrad_implicit_direct": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_wgrad_tn": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "ctta_wgrad_rowsum": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "ctta_wgrad_implicit": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "ctta_wgrad_scatter_rows": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ctta_wgrad_scatter_rows_bias": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_void_p, c_void_p, c_int, c_void_p]),
    "ctta_col_scatter": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ctta_transpose_multi": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "ctta_vae_encoder_create": (c_int, [POINTER(VAEConfig), POINTER(Tensor), c_int, c_void_p, POINTER(c_void_p)]),
    "ctta_vae_encoder_destroy": (None, [c_void_p]),
    "ctta_vae_encoder_load_weights": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ctta_vae_encode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_vae_encoder_num_taps": (c_int, [c_void_p]),
    "ctta_vae_encoder_tap_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int * 4)]),
    "ctta_vae_encoder_tap_read": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_mel_frontend_create": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, POINTER(c_void_p)]),
    "ctta_mel_frontend_destroy": (None, [c_void_p]),
    "ctta_wav_to_fbank": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_lincomb2_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_float, c_void_p]),
    "ctta_weighted_mse_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ctta_weighted_mse_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_pack_weight_multi": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "ctta_pack_weight_rows_multi": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_copy_segments_multi": (c_int, [c_void_p, c_int, c_void_p]),
    "ctta_unet_forward_train": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_unet_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ctta_unet_backward_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ctta_unet_backward_next": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_unet_arena_bytes": (c_size_t, [c_void_p]),
    "ctta_unet_num_taps": (c_int, [c_void_p]),
    "ctta_unet_tap_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int * 4)]),
    "ctta_unet_tap_read": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_vae_create": (c_int, [POINTER(VAEConfig), POINTER(Tensor), c_int, c_void_p, POINTER(c_void_p)]),
    "ctta_vae_destroy": (None, [c_void_p]),
    "ctta_t5_create": (c_int, [POINTER(T5Config), POINTER(Tensor), c_int, c_void_p, POINTER(c_void_p)]),
    "ctta_t5_destroy": (None, [c_void_p]),
    "ctta_t5_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_t5_arena_bytes": (c_size_t, [c_void_p]),
    "ctta_attention_rel": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ctta_vae_decode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_vae_decode_with_grad": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_vae_decode_backward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_vae_arena_bytes": (c_size_t, [c_void_p]),
    "ctta_vae_num_taps": (c_int, [c_void_p]),
    "ctta_vae_tap_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int * 4)]),
    "ctta_vae_tap_read": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_hifigan_create": (c_int, [POINTER(HifiganConfig), POINTER(Tensor), c_int, c_void_p, POINTER(c_void_p)]),
    "ctta_hifigan_destroy": (None, [c_void_p]),
    "ctta_hifigan_out_len": (c_int64, [c_void_p, c_int]),
    "ctta_hifigan_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_hifigan_forward_with_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_hifigan_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_wav_finalize": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctta_wav_extrema": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "ctta_wav_center": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctta_hifigan_arena_bytes": (c_size_t, [c_void_p]),
    "ctta_hifigan_num_taps": (c_int, [c_void_p]),
    "ctta_hifigan_tap_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int * 4)]),
    "ctta_hifigan_tap_read": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_heun_scale_model_input": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ctta_heun_add_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ctta_heun_step_first": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ctta_heun_step_second": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ctta_cfg_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ctta_snr_mse_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ctta_ema_update2": (c_int, [c_void_p, c_void_p, c_double, c_void_p, c_double, c_int64, c_void_p]),
    "ctta_conv_gemm": (c_int, [POINTER(ConvDesc), c_void_p]),
    "ctta_conv_last_gn_chunks": (c_int, []),
    "ctta_conv_bind_workspace": (None, [c_void_p, c_size_t]),
    "ctta_conv_bind_workspace_ex": (None, [c_void_p, c_size_t, c_int]),
    "ctta_conv_bound_workspace_header": (c_int, []),
    "ctta_conv_workspace_header_bytes": (c_size_t, []),
    "ctta_conv_bound_workspace": (None, [POINTER(c_void_p), POINTER(c_size_t)]),
    "ctta_conv_workspace_bytes": (c_size_t, []),
    "ctta_conv_suppress_splitk": (None, [c_int]),
    "ctta_conv_debug_stamps": (None, [c_void_p]),
    "ctta_attention_debug_stamps": (None, [c_void_p]),
    "ctta_conv_gemm_num_variants": (c_int, []),
    "ctta_conv_gemm_variant_name": (c_char_p, [c_int]),
    "ctta_attention_fullbias": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ctta_attention_fullbias_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                            c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                            c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                            c_int, c_int, c_int, c_float, c_void_p]),
    "ctta_resample_poly": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "ctta_resample_poly_bwd": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                       c_void_p]),
    "ctta_wav_to_logmel_db": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ctta_stft_create": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "ctta_stft_destroy": (None, [c_void_p]),
    "ctta_stft_frames": (c_int, [c_void_p, c_int]),
    "ctta_stft_magnitude": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_stft_magnitude_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctta_wav_to_logmel_db_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ctta_htsat_image": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_void_p, c_void_p]),
    "ctta_htsat_image_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_void_p, c_void_p]),
    "ctta_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "ctta_gelu": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ctta_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ctta_mean_tokens": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_mean_tokens_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_resunit_supported": (c_int, [c_int, c_int, c_int]),
    "ctta_ffn_desc_init": (None, [c_void_p]),
    "ctta_ffn_proj_pack_bytes": (c_size_t, [c_int, c_int]),
    "ctta_ffn_proj_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_ffn_block": (c_int, [c_void_p, c_void_p]),
    "ctta_ffn_geglu_supported": (c_int, [c_int, c_int]),
    "ctta_ffn_geglu_wanted": (c_int, [c_int, c_int, c_int64]),
    "ctta_ffn_pack_bytes": (c_size_t, [c_int, c_int]),
    "ctta_ffn_pack": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_ffn_geglu": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "ctta_frag_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_resunit_conv1d": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_float, c_void_p, c_int, c_float, c_float, c_void_p]),
    "ctta_logmel_to_image": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctta_avgpool2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_cnn14_head": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_reschain_supported": (c_int, [c_int, c_int, c_void_p]),
    "ctta_reschain_conv1d": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_float, c_void_p, c_int, c_float, c_float, c_void_p]),
    "ctta_conv_small_n": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_pack_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ctta_nhwc_f32_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_nhwc_bf16_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_rows_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "ctta_concat_channels": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "ctta_concat_channels_gn": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int64,
                                        c_void_p, c_void_p]),
    "ctta_groupnorm_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ctta_groupnorm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p]),
    "ctta_groupnorm_stats_out": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_groupnorm_from_partials": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int,
                                             c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_layernorm": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p]),
    "ctta_geglu": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "ctta_softmax_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "ctta_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ctta_linear_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_time_features": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ctta_fourier_features": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ctta_transpose_bf16": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "ctta_im2col_t": (c_int, [c_void_p] + [c_int] * 13 + [c_void_p, c_int, c_int, c_void_p]),
    "ctta_wgrad_scatter": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "ctta_row_scatter": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "ctta_groupnorm_bwd_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ctta_groupnorm_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ctta_groupnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ctta_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "ctta_layernorm_bwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "ctta_layernorm_bwd_scratch_floats": (c_size_t, [c_int64, c_int]),
    "ctta_layernorm_bwd_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ctta_geglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "ctta_add_slices": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "ctta_zero_insert2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_lrelu_bwd": (c_int, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ctta_conv_cout1_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_float, c_void_p, c_void_p]),
    "ctta_pool2_sum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_softmax_bias_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "ctta_softmax_bwd_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "ctta_linear_f32_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ctta_snr_mse_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ctta_adamw_ema2_zero": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_double, c_void_p, c_double, c_int,
                                     c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p]),
    "ctta_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p]),
    "ctta_prof_enable": (None, [c_int]),
    "ctta_set_option": (c_int, [c_char_p, c_int]),
    "ctta_get_option": (c_int, [c_char_p, POINTER(c_int)]),
    "ctta_num_options": (c_int, []),
    "ctta_option_name": (c_char_p, [c_int]),
    "ctta_option_default": (c_int, [c_int]),
    "ctta_set_gn_fuse": (None, [c_int]),
    "ctta_get_gn_fuse": (c_int, []),
    "ctta_prof_collect": (c_int, [c_int, POINTER(c_double), POINTER(c_double), POINTER(c_int64), c_char_p]),
}

_lib = None


def build(force=False):
    """Compiles csrc/*.hip for gfx950 into consistencytta_amd/libctta_hip.so (in-tree)."""
    if force:
        for f in os.listdir(os.path.join(CSRC, "build")) if os.path.isdir(os.path.join(CSRC, "build")) else []:
            os.remove(os.path.join(CSRC, "build", f))
    subprocess.run(["bash", os.path.join(CSRC, "build.sh")], check=True)
    return LIB_PATH


def lib():
    """Loads the library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libctta_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or consistencytta_amd/csrc/build.sh; there is no non-HIP fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64; import torch first so that this library binds to the SAME HIP runtime
    # instance (loading ours first pulls in /opt/rocm's copy: two runtimes in one process, device memory of one is
    # invisible to the other -- observed as "no ROCm-capable device is detected" from the second one)
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    # CTTA_OPT_<NAME>=<int> in the environment of the PYTHON process sets a library option at load time (A/B scripts:
    # tools/gpu_call.sh ab).  The library itself reads no environment variable; see ctta_set_option in include/ctta.h.
    for i in range(L.ctta_num_options()):
        name = L.ctta_option_name(i).decode()
        v = os.environ.get("CTTA_OPT_" + name.upper())
        if v is not None:
            set_option(name, int(v))
    return L


def set_option(name, value):
    """ctta_set_option (include/ctta.h): the library's only switches."""
    check(lib().ctta_set_option(name.encode(), int(value)))


def get_option(name):
    v = c_int()
    check(lib().ctta_get_option(name.encode(), ctypes.byref(v)))
    return v.value


def options():
    L = lib()
    return {L.ctta_option_name(i).decode(): get_option(L.ctta_option_name(i).decode()) for i in range(L.ctta_num_options())}


class CttaError(RuntimeError):
    pass


def check(status):
    if status != 0:
        msg = lib().ctta_last_error()
        raise CttaError("ctta status %d: %s" % (status, msg.decode() if msg else "?"))


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def tensor_table(sd):
    """dict name -> contiguous fp32 CUDA tensor  =>  (ctypes array of ctta_tensor, keepalive)."""
    import torch
    n = len(sd)
    arr = (Tensor * n)()
    keep = []
    for i, (k, v) in enumerate(sd.items()):
        if v.dtype != torch.float32 or not v.is_cuda:
            raise CttaError("weight '%s' must be a float32 CUDA tensor (got %s on %s)" % (k, v.dtype, v.device))
        v = v.contiguous()
        kb = k.encode()
        keep.append((kb, v))
        arr[i].name = kb
        arr[i].data = v.data_ptr()
        arr[i].ndim = v.ndim
        for d in range(v.ndim):
            arr[i].shape[d] = v.shape[d]
    return arr, keep
