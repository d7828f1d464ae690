"""ctypes binding of libdxmi_hip.so (the C-ABI declared in include/dxmi_hip.h).

The library is the ONLY compute path of this package: if it is missing, or no gfx950 device is
visible when a kernel is requested, we raise — there is no CPU or torch fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DXMI_LIB") or os.path.join(_HERE, "libdxmi_hip.so")      # DXMI_LIB: another build of the same library (A/B timing)

c_void_p, c_int, c_float, c_int64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_int64


class ConvDesc(ctypes.Structure):
    """Mirror of struct dxmi_conv_desc (include/dxmi_hip.h)."""

    _fields_ = [
        ("in0", c_void_p), ("in1", c_void_p), ("wpacked", c_void_p), ("bias", c_void_p),
        ("addvec", c_void_p), ("residual", c_void_p), ("out", c_void_p),
        ("N", c_int), ("IH", c_int), ("IW", c_int), ("C0", c_int), ("C1", c_int),
        ("OH", c_int), ("OW", c_int), ("Cout", c_int),
        ("ksize", c_int), ("stride", c_int), ("pad", c_int), ("upsample", c_int), ("act", c_int),
        ("addvec_ld", c_int), ("in_mode", c_int), ("out_mode", c_int), ("variant", c_int),
        ("mask_src", c_void_p), ("mask_slope", c_float), ("gn_stats", c_void_p),
        ("gn_out", c_void_p), ("gn_gamma", c_void_p), ("gn_beta", c_void_p), ("gn_eps", c_float), ("gn_groups", c_int), ("gn_flags", c_int),
    ]


class PackItem(ctypes.Structure):
    """Mirror of struct dxmi_pack_item (include/dxmi_hip.h)."""

    _fields_ = [("w", c_void_p), ("dst", c_void_p), ("Cout", c_int), ("Cin", c_int), ("ksize", c_int), ("transpose_flip", c_int),
                ("k27", c_int)]


# symbol -> (restype, argtypes); kept in one table so tests can check it against the header.
SIGNATURES = {
    "dxmi_last_error": (ctypes.c_char_p, []),
    "dxmi_version": (c_int, []),
    "dxmi_device_check": (c_int, []),
    "dxmi_set_tuning": (c_int, [ctypes.c_char_p, c_int]),
    "dxmi_get_tuning": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int)]),
    "dxmi_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p]),
    "dxmi_conv2d_kernel_id": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dxmi_conv2d_gn_stats_partials": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dxmi_conv2d_gn_fuse_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dxmi_conv_ws_last_clock": (c_int, [ctypes.POINTER(ctypes.c_uint64)]),
    "dxmi_gn_block_stats_partials": (c_int, [c_int]),
    "dxmi_gn_block_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_gn_stats_fold": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dxmi_groupnorm_apply": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dxmi_groupnorm_apply_split": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                           c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dxmi_gn_blockstats_to_generic": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_attention_fwd_lse_supported": (c_int, [c_int, c_int, c_int]),
    "dxmi_attention_fwd_lse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dxmi_attention_bwd_lse": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dxmi_gn_ss_grads": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dxmi_fid_stats_workspace_bytes": (c_int64, [c_int64, c_int]),
    "dxmi_fid_stats": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dxmi_packed_conv_weight_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "dxmi_pack_conv_weights": (c_int, [ctypes.POINTER(PackItem), c_int, c_void_p]),
    "dxmi_pack_conv_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dxmi_conv2d_wgrad_workspace_bytes": (c_int64, [c_int] * 6),
    "dxmi_conv2d_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "dxmi_conv2d_wgrad_bias": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "dxmi_groupnorm_silu_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 9 + [c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dxmi_bgemm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "dxmi_softmax_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "dxmi_attention_bwd_supported": (c_int, [c_int, c_int, c_int]),
    "dxmi_attention_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "dxmi_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dxmi_colsum_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "dxmi_colsum_blocks_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "dxmi_colsum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_pool_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dxmi_value_head_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_groupnorm_silu_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dxmi_attention_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dxmi_attention_proj_supported": (c_int, [c_int, c_int, c_int]),
    "dxmi_pack_attn_proj_weight": (c_int, [c_void_p, c_void_p, c_void_p]),
    "dxmi_attention_proj_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dxmi_attn_block_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "dxmi_attn_block_packed_bytes": (c_int64, []),
    "dxmi_attn_block_pack": (c_int, [c_void_p] * 7 + [c_float, c_void_p, c_void_p]),
    "dxmi_attn_block_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_timestep_embedding": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "dxmi_linear_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dxmi_linear_splitk_slices": (c_int, [c_int, c_int, c_int]),
    "dxmi_linear_splitk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dxmi_var_step_fwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_void_p]),
    "dxmi_var_step_bwd": (c_int, [c_void_p] * 9 + [c_int, c_int, c_void_p]),
    "dxmi_edm_step_bwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_float, c_void_p]),
    "dxmi_var_gather_sched": (c_int, [c_void_p] * 9 + [c_int, c_int, c_void_p]),
    "dxmi_pool_act": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dxmi_value_head": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_groupnorm_silu_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "dxmi_groupnorm_silu_bwd_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "dxmi_groupnorm_generic_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "dxmi_groupnorm_generic_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dxmi_groupnorm_generic_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "dxmi_groupnorm_generic_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dxmi_groupnorm_generic_bwd_saved": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dxmi_upsample2x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dxmi_edm_precond": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "dxmi_edm_step_fwd": (c_int, [c_void_p] * 8 + [c_int, c_int, c_float, c_void_p]),
    "dxmi_quantize_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dxmi_im2col27": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_nhwc_bf16_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_mt_blocks": (c_int64, [c_void_p, c_int]),
    "dxmi_adam_step": (c_int, [c_void_p] * 6 + [c_int] + [ctypes.c_double] * 4 + [c_void_p, c_int, c_void_p]),
    "dxmi_adam_step_dev": (c_int, [c_void_p] * 5 + [c_int] + [ctypes.c_double] * 3 + [c_void_p, c_void_p, c_int, c_void_p]),
    "dxmi_radam_step_dev": (c_int, [c_void_p] * 5 + [c_int] + [ctypes.c_double] * 3 + [c_void_p, c_void_p, c_void_p, c_void_p]),
    "dxmi_radam_step": (c_int, [c_void_p] * 6 + [c_int] + [ctypes.c_double] * 6 + [c_void_p, c_void_p, c_void_p]),
    "dxmi_gradnorm_clip": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "dxmi_dropout_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_float, ctypes.c_uint32, c_void_p]),
    "dxmi_dropout_bf16_dev": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "dxmi_td_gather_cost": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int64, c_void_p]),
    "dxmi_td_loss": (c_int, [c_void_p] * 5 + [c_int, c_void_p]),
    "dxmi_value_head_pgrad": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "dxmi_gconv_packed_elems": (c_int64, [c_int] * 4),
    "dxmi_gconv_pack": (c_int, [c_void_p] * 5 + [c_float, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "dxmi_gconv_fwd": (c_int, [c_void_p] * 4 + [c_int] * 14 + [c_void_p]),
    "dxmi_pool3x3": (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "dxmi_global_avgpool": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dxmi_resize_bilinear_nhwc16": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "dxmi_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
}

_lib = None


class DxmiError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DxmiError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C diffusion-by-maxentirl_amd/csrc`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().dxmi_last_error().decode("utf-8", "replace")
        raise DxmiError(f"{what} failed ({status}): {msg}")
