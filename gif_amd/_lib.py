"""ctypes binding of libgif_hip.so (the C ABI declared in include/gif_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C gif_amd/csrc`.
"""
import ctypes
import os

# torch FIRST: it ships its own libamdhip64 and must be the copy the process binds — when this library (linked against the ROCm
# install's runtime) is loaded before torch, the process ends up with two HIP runtimes and every launch from here fails with "no
# ROCm-capable device is detected" (found in round 6: __graft_entry__.build() followed by smoke() in ONE process)
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgif_hip.so")

c_int, c_i64, c_float, c_void_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


def knob(name, default):
    """Value of an A/B / ablation / probe environment knob.  Such GIF_* variables (they select measured-slower, diagnostic or
    deliberately wrong paths) are honoured ONLY in a process that opted in with GIF_EXPERIMENTAL=1, so that a stray variable in a
    production environment changes neither dispatch nor numerics; otherwise the default is returned (csrc/common.h gif::knob is
    the same gate for the knobs the library reads).  NOT gated: GIF_FP32_MFMA (the documented contraction-mode switch),
    GIF_PROF_DUMP (profiling output), GIF_FORCE_DIST (test hook: collective code paths in a one-rank group; changes no arithmetic)."""
    if os.environ.get("GIF_EXPERIMENTAL", "0") in ("", "0"):
        return default
    return os.environ.get(name, default)
P = c_void_p  # every device pointer travels as an integer address


class ConvGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "Hb", "Wb", "Cb", "Hs", "Ws", "Cs", "KH", "KW", "stride", "pad")]


class ConvEpilogue(ctypes.Structure):
    # ABI 2: + the gradient-producer fusions (mask of the leaky ReLU the gradient flows into, modulation-gradient dot product,
    # bias-gradient column sums; include/gif_hip.h)
    _fields_ = [("in_scale", P), ("out_scale", P), ("bias", P), ("residual", P),
                ("act", ctypes.c_int32), ("slope", c_float), ("gain", c_float),
                ("mask_src", P), ("mask_slope", c_float), ("mask_gain", c_float),
                ("dot_src", P), ("dot", P), ("colsum", P), ("red_ws", P), ("out_f32", ctypes.c_int32)]


class LinearBankSeg(ctypes.Structure):
    # one layer of the modulation bank (gif_linear_bank_seg, include/gif_hip.h)
    _fields_ = [("w", P), ("bias", P), ("s", P), ("gs", P), ("gw", P), ("gbias", P), ("n", ctypes.c_int32), ("reserved", ctypes.c_int32)]


GP, EP = ctypes.POINTER(ConvGeom), ctypes.POINTER(ConvEpilogue)

# name -> (restype, argtypes); must list every symbol of include/gif_hip.h (tests/test_abi.py checks it)
PROTOTYPES = {
    "gif_last_error": (ctypes.c_char_p, []),
    "gif_abi_version": (c_int, []),
    "gif_f16_overflow_clear": (c_int, [P]),
    "gif_f16_overflow_or_into": (c_int, [P, P]),
    "gif_f16_overflow_watch": (c_int, [c_int]),
    "gif_set_fp32_mfma_mode": (c_int, [c_int]),
    "gif_get_fp32_mfma_mode": (c_int, []),
    "gif_rasterize_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "gif_rasterize_assume_clean_workspace": (c_int, [P, c_int]),
    "gif_rasterize_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "gif_rasterize_colors_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "gif_rasterize_f64": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "gif_rasterize_colors_f64": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "gif_vertex_normals_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "gif_texture_map_f32": (c_int, [P] * 9 + [c_int] * 6 + [P]),
    "gif_texture_map_bwd_f32": (c_int, [P] * 8 + [c_int] * 6 + [P]),
    "gif_conv_epilogue_ws_floats": (c_i64, [c_i64, c_int]),
    "gif_conv2d_pack_dims": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gif_pack_weight_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_conv2d_fwd_f32": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_bwd_data_f32": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_fwd_f32x3": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_bwd_data_f32x3": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_x3_eligible": (c_int, [c_int, c_int]),
    "gif_pack_weight_f32h2_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "gif_pack_weight_f32h2": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_pack_weight_f32h2x3": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_pack_weight_f32h2_tapdense_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "gif_pack_weight_f32h2x3_tapdense": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_conv2d_fwd_f32h2_tapdense": (c_int, [P, P, P, P, GP, EP, P]),
    "gif_conv2d_bwd_data_f32h2_tapdense": (c_int, [P, P, P, P, GP, EP, P]),
    "gif_conv2d_fwd_f32h2": (c_int, [P, P, P, P, GP, EP, P]),
    "gif_conv2d_bwd_data_f32h2": (c_int, [P, P, P, P, GP, EP, P]),
    "gif_h2_fallback_stats": (c_int, [ctypes.POINTER(ctypes.c_uint64), c_int]),
    "gif_conv2d_fwd_f32x3_tapdense": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_bwd_data_f32x3_tapdense": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_x3_tapdense_steps": (c_int, [c_int, c_int, c_int]),
    "gif_pack_weight_f32x3_tapdense": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_conv2d_pack_dims_x3": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gif_pack_weight_f32x3": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_conv2d_wgrad_dims": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gif_conv2d_wgrad_splits": (c_int, [GP]),
    "gif_conv2d_wgrad_f32": (c_int, [P, P, P, P, P, GP, c_int, P]),
    "gif_conv2d_wgrad_f32x3": (c_int, [P, P, P, P, P, GP, c_int, P]),
    "gif_conv2d_wgrad_f32h2": (c_int, [P, P, P, P, P, GP, c_int, P]),
    "gif_unpack_wgrad_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_winograd_pack_dims": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gif_winograd_pack_dims_x3": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gif_winograd_workspace_floats": (c_i64, [c_int, c_int, c_int, c_int]),
    "gif_winograd_weight_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_int, c_float, P]),
    "gif_winograd_weight_f32x3": (c_int, [P, P, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_int, c_float, P]),
    "gif_conv3x3_winograd_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, EP, P]),
    "gif_conv3x3_winograd_f32x3": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, EP, P]),
    "gif_winograd_weight_f32h2_bytes": (c_i64, [c_int, c_int]),
    "gif_winograd_weight_f32h2": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_int, c_float, P]),
    "gif_conv3x3_winograd_f32h2": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, EP, P]),
    "gif_conv3x3_winograd_wgrad_splits": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "gif_conv3x3_winograd_wgrad_f32": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_conv3x3_winograd_wgrad_f32x3": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_conv3x3_winograd_wgrad_f32h2": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_winograd_unpack_wgrad_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_texture_pair_loss_partials": (c_int, [c_i64]),
    "gif_texture_pair_loss_f32": (c_int, [P, P, P, P, P, P, P, c_int, c_i64, P]),
    "gif_texture_pair_loss_bwd_f32": (c_int, [P, P, P, P, P, P, P, c_int, c_i64, P]),
    "gif_resize_f32": (c_int, [P, P, c_i64, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_resize_bwd_f32": (c_int, [P, P, c_i64, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_upfirdn2d_f32": (c_int, [P, P, P] + [c_int] * 13 + [EP, P]),
    "gif_bias_act_f32": (c_int, [P, P, P, P, c_i64, c_int, c_float, c_float, P]),
    "gif_colsum_partial_floats": (c_i64, [c_i64, c_int]),
    "gif_bias_act_bwd_f32": (c_int, [P, P, P, P, P, c_i64, c_int, c_float, c_float, P]),
    "gif_colsum_f32": (c_int, [P, P, P, c_i64, c_int, P]),
    "gif_mul_reduce_chunks": (c_int, [c_i64]),
    "gif_mul_reduce_f32": (c_int, [P, P, P, P, P, P, c_int, c_i64, c_int, P]),
    "gif_pack_nhwc_f32": (c_int, [P, c_int, c_int, ctypes.POINTER(c_i64), P, c_int, c_int, ctypes.POINTER(c_i64), P, c_int, c_int, c_int, c_int, P]),
    "gif_pack_nhwc_f16": (c_int, [P, c_int, c_int, ctypes.POINTER(c_i64), P, c_int, c_int, ctypes.POINTER(c_i64), P, c_int, c_int, c_int, c_int, P]),
    "gif_unpack_nhwc_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_unpack_nhwc_f16": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_bilinear_down_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_act_inv_mul_reduce_f32": (c_int, [P, P, P, P, P, P, c_int, c_i64, c_int, c_float, c_float, P]),
    "gif_mbstd_fwd_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_mbstd_bwd_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "gif_sqnorm_per_sample_f32": (c_int, [P, P, c_int, c_i64, P]),
    "gif_conv2d_pack_dims_f16": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gif_pack_weight_f16": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_float, P]),
    "gif_conv2d_f16_halo_eligible": (c_int, [c_int] * 7),
    "gif_conv2d_f16_halo_enable": (c_int, [c_int]),
    "gif_conv2d_fwd_f16": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_bwd_data_f16": (c_int, [P, P, P, GP, EP, P]),
    "gif_conv2d_wgrad_dims_f16": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gif_conv2d_wgrad_splits_f16": (c_int, [GP]),
    "gif_conv2d_wgrad_f16": (c_int, [P, P, P, P, P, GP, c_int, P]),
    "gif_upfirdn2d_f16": (c_int, [P, P, P] + [c_int] * 13 + [EP, P]),
    "gif_bias_act_f16": (c_int, [P, P, P, P, c_i64, c_int, c_float, c_float, P]),
    "gif_bias_act_bwd_f16": (c_int, [P, P, P, P, P, c_i64, c_int, c_float, c_float, P]),
    "gif_colsum_f16": (c_int, [P, P, P, c_i64, c_int, P]),
    "gif_mul_reduce_f16": (c_int, [P, P, P, P, P, P, c_int, c_i64, c_int, P]),
    "gif_act_inv_mul_reduce_f16": (c_int, [P, P, P, P, P, P, c_int, c_i64, c_int, c_float, c_float, P]),
    "gif_linear_nt_f32": (c_int, [P, P, P, P] + [c_int] * 7 + [c_float, c_int, c_float, c_float, P]),
    "gif_linear_nn_f32": (c_int, [P, P, P] + [c_int] * 7 + [c_float, P]),
    "gif_linear_tn_f32": (c_int, [P, P, P] + [c_int] * 6 + [c_float, P]),
    "gif_linear_bank_fwd_f32": (c_int, [P, c_int, c_int, c_int, ctypes.POINTER(LinearBankSeg), c_int, c_float, P]),
    "gif_linear_bank_bwd_f32": (c_int, [P, c_int, c_int, c_int, ctypes.POINTER(LinearBankSeg), c_int, c_float, P, c_int, c_int, P]),
    "gif_weight_sq_sum_f32": (c_int, [P, P, c_int, c_int, c_int, P]),
    "gif_style_demod_f32": (c_int, [P, P, P] + [c_int] * 7 + [c_float, c_float, P]),
    "gif_style_demod_bwd_s_f32": (c_int, [P] * 6 + [c_int] * 7 + [c_float, P]),
    "gif_style_demod_bwd_w_f32": (c_int, [P] * 4 + [c_int] * 5 + [c_float, P]),
    "gif_demod_wgrad_f32": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "gif_adam_chunk_floats": (c_int, []),
    "gif_adam_ema_step_f32": (c_int, [P, c_int, P, P, P, c_float, c_float, c_float, c_float, ctypes.c_double, ctypes.c_double,
                                      c_float, c_int, P, P, P, P]),
    "gif_prof_enable": (c_int, [c_int]),
    "gif_prof_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
}

_lib = None


class GifHipError(RuntimeError):
    pass


def load():
    """Load libgif_hip.so once; raise loudly when it is not built (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GifHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C gif_amd/csrc` "
            "(or __graft_entry__.build()). gif_amd has no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gif_last_error().decode(errors="replace")
        raise GifHipError(f"{what} failed (rc={rc}): {msg}")
