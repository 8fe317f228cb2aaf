"""ctypes binding of libfastgen_amd.so (C ABI: include/fastgen_amd.h).  There is no fallback: if the library is
missing or a call fails, this raises."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfastgen_amd.so")

FG_MAX_LEVELS = 8
FG_DTYPE_F32, FG_DTYPE_BF16, FG_DTYPE_BF16X3 = 0, 1, 2
DTYPE_NAMES = {"fp32": FG_DTYPE_F32, "bf16": FG_DTYPE_BF16, "bf16x3": FG_DTYPE_BF16X3}
FG_SAMPLE_SDE, FG_SAMPLE_ODE = 0, 1
FG_LOOP_X0, FG_LOOP_MEANFLOW, FG_LOOP_EULER = 0, 1, 2
FG_SCHEDULE_EDM, FG_SCHEDULE_RF = 0, 1
FG_DROP_PRECOND_INPUT, FG_DROP_PRECOND_OUTPUT = 1, 2
FG_BWD_DECODER, FG_BWD_ENCODER, FG_BWD_EMBED = 1, 2, 4


class fg_edm_config(ctypes.Structure):
    _fields_ = [
        ("img_resolution", c_int), ("img_channels", c_int), ("label_dim", c_int), ("augment_dim", c_int),
        ("model_channels", c_int), ("num_levels", c_int), ("channel_mult", c_int * FG_MAX_LEVELS),
        ("channel_mult_emb", c_int), ("num_blocks", c_int), ("num_attn_resolutions", c_int),
        ("attn_resolutions", c_int * FG_MAX_LEVELS), ("channel_mult_noise", c_int), ("sigma_data", c_double),
        ("sigma_shift", c_double), ("compute_dtype", c_int), ("r_timestep", c_int), ("drop_precond", c_int),
        ("schedule", c_int),
    ]


class fg_dit_config(ctypes.Structure):
    _fields_ = [("input_size", c_int), ("patch_size", c_int), ("in_channels", c_int), ("hidden_size", c_int), ("depth", c_int),
                ("num_heads", c_int), ("mlp_hidden", c_int), ("embedding_rows", c_int), ("r_timestep", c_int), ("compute_dtype", c_int)]


class fg_wan_config(ctypes.Structure):
    _fields_ = [("num_heads", c_int), ("head_dim", c_int), ("in_channels", c_int), ("out_channels", c_int), ("text_dim", c_int),
                ("freq_dim", c_int), ("ffn_dim", c_int), ("num_layers", c_int), ("rope_max_seq_len", c_int), ("chunk_size", c_int),
                ("total_num_frames", c_int), ("eps", c_float)]


class fg_dit_sampler_config(ctypes.Structure):
    _fields_ = [("t_scale", c_double), ("guidance_scale", c_double), ("use_sit_convention", c_int), ("time_cond_diff", c_int),
                ("net_pred_flow", c_int), ("schedule", c_int)]


class fg_wan_sampler_config(ctypes.Structure):
    _fields_ = [("t_scale", c_double), ("context_noise", c_double), ("net_pred_flow", c_int), ("schedule", c_int), ("prefill_frames", c_int)]


# name -> (restype, argtypes); every symbol include/fastgen_amd.h declares
SIGNATURES = {
    "fg_last_error": (c_char_p, []),
    "fg_version": (c_char_p, []),
    "fg_edm_create": (c_int, [POINTER(fg_edm_config), POINTER(c_void_p)]),
    "fg_edm_destroy": (None, [c_void_p]),
    "fg_edm_num_params": (c_int, [c_void_p]),
    "fg_edm_param_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int64)]),
    "fg_edm_bind_param": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "fg_edm_pack_weights": (c_int, [c_void_p, c_void_p]),
    "fg_edm_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "fg_edm_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                               c_size_t, c_void_p]),
    "fg_edm_num_feature_taps": (c_int, [c_void_p]),
    "fg_edm_feature_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int)]),
    "fg_edm_forward_features": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p), c_int,
                                        c_void_p, c_size_t, c_void_p]),
    "fg_sampler_run": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double), c_int, c_int, c_int, c_void_p, c_uint64,
                               c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "fg_edm_t_list": (c_int, [c_int, POINTER(c_double)]),
    "fg_rf_t_list": (c_int, [c_int, POINTER(c_double)]),
    "fg_edm_profile_begin": (c_int, [c_void_p]),
    "fg_edm_profile_end": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    "fg_edm_num_blocks": (c_int, [c_void_p]),
    "fg_edm_block_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                  POINTER(c_int), POINTER(c_int)]),
    "fg_edm_run_block": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                 c_size_t, c_void_p]),
    "fg_op_gn_coeffs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "fg_op_latents": (c_int, [c_void_p, c_double, c_void_p, c_int64, c_void_p]),
    "fg_op_forward_process": (c_int, [c_void_p, c_void_p, c_double, c_int, c_void_p, c_int64, c_void_p]),
    "fg_op_x0_to_eps": (c_int, [c_void_p, c_void_p, c_double, c_int, c_void_p, c_int64, c_void_p]),
    "fg_op_conv_wgrad_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "fg_op_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    "fg_op_conv_wgrad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    "fg_edm_bind_grad": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "fg_edm_block_backward_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "fg_edm_run_block_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "fg_edm_backward_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "fg_edm_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                c_size_t, c_void_p]),
    "fg_edm_forward_train": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_size_t, c_void_p]),
    "fg_edm_backward_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_void_p, c_size_t, c_void_p]),
    "fg_edm_backward_part": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fg_dit_create": (c_int, [POINTER(fg_dit_config), POINTER(c_void_p)]),
    "fg_dit_destroy": (None, [c_void_p]),
    "fg_dit_num_params": (c_int, [c_void_p]),
    "fg_dit_param_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int64)]),
    "fg_dit_bind_param": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "fg_dit_pack_weights": (c_int, [c_void_p, c_void_p]),
    "fg_dit_pack_group": (c_int, [c_void_p, c_char_p, c_char_p, c_void_p]),
    "fg_dit_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "fg_dit_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "fg_dit_forward_features": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int), POINTER(c_void_p),
                                        c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fg_dit_sampler_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "fg_dit_sampler_run": (c_int, [c_void_p, POINTER(fg_dit_sampler_config), c_void_p, c_void_p, c_void_p, POINTER(c_double), c_int, c_int, c_int,
                                   c_void_p, c_uint64, c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "fg_wan_sampler_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int]),
    "fg_wan_sampler_run": (c_int, [c_void_p, POINTER(fg_wan_sampler_config), c_void_p, POINTER(c_double), c_int, c_int, POINTER(c_int), c_void_p,
                                   c_uint64, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "fg_wan_create": (c_int, [POINTER(fg_wan_config), POINTER(c_void_p)]),
    "fg_wan_destroy": (None, [c_void_p]),
    "fg_wan_num_params": (c_int, [c_void_p]),
    "fg_wan_param_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int64)]),
    "fg_wan_bind_param": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "fg_wan_pack_weights": (c_int, [c_void_p, c_void_p]),
    "fg_wan_pack_group": (c_int, [c_void_p, c_char_p, c_char_p, c_void_p]),
    "fg_wan_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int]),
    "fg_wan_clear_caches": (c_int, [c_void_p, c_void_p]),
    "fg_wan_set_text": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fg_wan_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                               c_void_p]),
    "fg_wan_forward_block_causal": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fg_disc_edm_num_params": (c_int, [c_int]),
    "fg_disc_edm_workspace_bytes": (c_size_t, [c_int, c_int]),
    "fg_disc_edm_run": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "fg_edm_jvp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                           c_void_p, c_size_t, c_void_p]),
    "fg_edm_set_dropout": (c_int, [c_void_p, c_float, c_uint64]),
    "fg_op_dropout_mask": (c_int, [c_void_p, c_int64, c_float, ctypes.c_uint32, c_uint64, c_void_p]),
    "fg_edm_set_augment": (c_int, [c_void_p, c_void_p]),
    "fg_edm_set_training": (c_int, [c_void_p, c_int]),
    "fg_op_images_to_u8": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "fg_op_randn": (c_int, [c_void_p, c_int64, c_uint64, c_uint64, c_void_p]),
    "fg_op_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fg_op_attention_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fg_op_gemm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                c_int, c_void_p]),
}

_lib = None


def lib():
    """Load (once) and return the library.  Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C fastgen_amd/csrc`).  fastgen_amd has no non-HIP fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class FastGenAMDError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        msg = lib().fg_last_error().decode("utf-8", "replace")
        raise FastGenAMDError(f"fastgen_amd error {rc}: {msg}")
