"""ctypes binding of libcasmvs_hip.so (C ABI declared in include/casmvs.h).

This is the stub a maintainer of the reference would add to call the MI355X engine from
`models/modules.py` / `models/mvsnet.py`.  There is no CPU fallback: if the
library is missing the import of any op raises, and on a machine without a gfx950 device the
launches return CASMVS_ERR_HIP which is raised as RuntimeError.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_void_p

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# CASMVS_LIB_PATH: load another BUILD of the same library (profiling: -DCASMVS_TRACE, compiler-flag A/B runs)
LIB_PATH = os.environ.get("CASMVS_LIB_PATH") or os.path.join(_PKG_DIR, "libcasmvs_hip.so")
ABI_VERSION = 6

CONV_S1, CONV_S2, CONV_T2 = 0, 1, 2
CONV2D_K3, CONV2D_K5S2, CONV2D_K1, CONV2D_K1_UP = 3, 4, 5, 6

# every symbol include/casmvs.h declares: name -> (restype, argtypes)
_FP = c_void_p  # device / host float* passed as integer addresses
SYMBOLS = {
    "casmvs_abi_version": (c_int, []),
    "casmvs_packed_opsel_safe": (c_int, []),
    "casmvs_last_error": (c_char_p, []),
    "casmvs_depth_hypotheses_f32": (c_int, [_FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_homo_warp_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_var_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_gwc_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_nchw_to_nhwc_f32": (c_int, [_FP, _FP, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_var_nhwc_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_gwc_nhwc_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_lds_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "casmvs_costvol_lds_preferred": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "casmvs_costvol_var_lds_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_gwc_lds_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_homo_warp_nhwc_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_homo_warp_lds_supported": (c_int, [c_int, c_int, c_int]),
    "casmvs_homo_warp_lds_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_partial_var_f32": (c_int, [_FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_partial_gwc_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_var_finalize_f32": (c_int, [_FP, _FP, _FP, c_size_t, c_int, c_void_p]),
    "casmvs_costvol_gwc_finalize_f32": (c_int, [_FP, _FP, c_size_t, c_int, c_void_p]),
    "casmvs_conv3d_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "casmvs_conv3d_pack_f32": (c_int, [c_int, c_int, c_int, _FP, _FP, _FP, _FP]),
    "casmvs_conv3d_forward_f32": (c_int, [c_int, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_conv0_splitbf16_packed_bytes": (c_size_t, [c_int]),
    "casmvs_conv0_splitbf16_pack": (c_int, [c_int, _FP, _FP, _FP, c_void_p]),
    "casmvs_conv0_splitbf16_supported": (c_int, [c_int, c_int]),
    "casmvs_conv0_splitbf16_forward_f32": (c_int, [c_void_p, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "casmvs_selftest_mfma_bf16": (c_int, [_FP]),
    "casmvs_conv0_splitf16_packed_bytes": (c_size_t, [c_int]),
    "casmvs_conv0_splitf16_pack": (c_int, [c_int, _FP, _FP, _FP, c_void_p]),
    "casmvs_conv0_splitf16_supported": (c_int, [c_int, c_int]),
    "casmvs_conv0_splitf16_forward_f32": (c_int, [c_void_p, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "casmvs_selftest_mfma_f16": (c_int, [_FP]),
    "casmvs_deconv9_splitf16_packed_bytes": (c_size_t, []),
    "casmvs_deconv9_splitf16_pack": (c_int, [_FP, _FP, _FP, c_void_p]),
    "casmvs_deconv9_splitf16_supported": (c_int, [c_int]),
    "casmvs_deconv9_splitf16_forward_f32": (c_int, [c_void_p, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_deconv11_splitf16_packed_bytes": (c_size_t, []),
    "casmvs_deconv11_splitf16_pack": (c_int, [_FP, _FP, _FP, c_void_p]),
    "casmvs_deconv11_splitf16_supported": (c_int, [c_int]),
    "casmvs_deconv11_splitf16_forward_f32": (c_int, [c_void_p, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_conv0_zmarch_supported": (c_int, [c_int, c_int]),
    "casmvs_conv0_zmarch_forward_f32": (c_int, [c_void_p, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_conv_ci_splitf16_packed_bytes": (c_size_t, [c_int, c_int]),
    "casmvs_conv_ci_splitf16_pack": (c_int, [c_int, c_int, _FP, _FP, _FP, c_void_p]),
    "casmvs_conv_ci_splitf16_supported": (c_int, [c_int, c_int, c_int]),
    "casmvs_conv_ci_splitf16_forward_f32": (c_int, [c_void_p, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_conv2d_k5s2_splitf16_packed_bytes": (c_size_t, [c_int, c_int]),
    "casmvs_conv2d_k5s2_splitf16_pack": (c_int, [c_int, c_int, _FP, _FP, _FP, c_void_p]),
    "casmvs_conv2d_k5s2_splitf16_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "casmvs_conv2d_k5s2_splitf16_forward_f32": (c_int, [c_void_p, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_conv11_prob_zfused_supported": (c_int, [c_int, c_int, c_int]),
    "casmvs_conv11_prob_zfused_f32": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "casmvs_conv_s2_splitf16_packed_bytes": (c_size_t, [c_int, c_int]),
    "casmvs_conv_s2_splitf16_pack": (c_int, [c_int, c_int, _FP, _FP, _FP, c_void_p]),
    "casmvs_conv_s2_splitf16_supported": (c_int, [c_int, c_int, c_int]),
    "casmvs_conv_s2_splitf16_forward_f32": (c_int, [c_void_p, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_costreg_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "casmvs_costreg_packed_floats": (c_size_t, [c_int, c_void_p]),
    "casmvs_costreg_pack_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "casmvs_costreg_forward_f32": (c_int, [POINTER(c_void_p), _FP, _FP, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, POINTER(c_void_p), c_void_p]),
    "casmvs_conv2d_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "casmvs_conv2d_pack_f32": (c_int, [c_int, c_int, c_int, _FP, _FP, _FP, _FP]),
    "casmvs_conv2d_forward_f32": (c_int, [c_int, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_featurenet_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "casmvs_featurenet_forward_f32": (c_int, [POINTER(c_void_p), _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_float, POINTER(c_void_p), c_void_p]),
    "casmvs_fpn_tail0_supported": (c_int, [c_int, c_int]),
    "casmvs_fpn_tail0_f32": (c_int, [_FP] * 6 + [c_int, c_int, c_int, c_void_p]),
    "casmvs_fnet_conv0_mm_packed_bytes": (c_size_t, []),
    "casmvs_fnet_conv0_mm_pack": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "casmvs_fnet_conv0_mm_supported": (c_int, [c_int]),
    "casmvs_fnet_conv0_mm_f32": (c_int, [c_void_p, _FP, _FP, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_fpn_tail0_splitf16_packed_bytes": (c_size_t, []),
    "casmvs_fpn_tail0_splitf16_pack": (c_int, [_FP, c_void_p]),
    "casmvs_fpn_tail0_splitf16_f32": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "casmvs_conv2d_ci_splitf16_packed_bytes": (c_size_t, [c_int, c_int]),
    "casmvs_conv2d_ci_splitf16_pack": (c_int, [c_int, c_int, _FP, _FP, _FP, c_void_p]),
    "casmvs_conv2d_ci_splitf16_supported": (c_int, [c_int, c_int, c_int]),
    "casmvs_conv2d_ci_splitf16_forward_f32": (c_int, [c_void_p, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "casmvs_featurenet_forward_fused_f32": (c_int, [POINTER(c_void_p), c_void_p, c_int, _FP, POINTER(c_void_p), _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_float, POINTER(c_void_p), c_void_p]),
    "casmvs_softmax_regress_f32": (c_int, [_FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_prob_regress_supported": (c_int, [c_int, c_int]),
    "casmvs_prob_regress_f32": (c_int, [_FP] * 7 + [c_int] * 5 + [c_float, c_int, c_void_p]),
    "casmvs_costreg_regress_f32": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, POINTER(c_void_p), c_void_p]),
    "casmvs_depth_regression_f32": (c_int, [_FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_fuse_reference_view": (c_int, [_FP] * 16 + [c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "casmvs_fuse_reference_view_paired": (c_int, [_FP] * 16 + [c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "casmvs_homo_warp_backward_workspace_bytes": (c_size_t, [c_int] * 5),
    "casmvs_homo_warp_backward_f32": (c_int, [_FP, _FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_softmax_regress_backward_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_conv_wgrad_workspace_bytes": (c_size_t, [c_int] * 7),
    "casmvs_conv_wgrad_f32": (c_int, [c_int, _FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_prob_wgrad_supported": (c_int, [c_int] * 4),
    "casmvs_prob_wgrad_workspace_bytes": (c_size_t, [c_int] * 4),
    "casmvs_prob_wgrad_f32": (c_int, [_FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_conv_dgrad_direct_f32": (c_int, [c_int, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_channel_sums_blocks": (c_int, [c_int, c_size_t]),
    "casmvs_channel_sums_f64": (c_int, [_FP, _FP, c_int, c_int, c_size_t, c_void_p]),
    "casmvs_abn_apply_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_size_t, c_float, c_void_p]),
    "casmvs_abn_backward_sums_f64": (c_int, [_FP] * 6 + [c_int, c_int, c_size_t, c_float, c_void_p]),
    "casmvs_abn_backward_apply_f32": (c_int, [_FP] * 9 + [c_int, c_int, c_size_t, c_float, c_void_p]),
    "casmvs_pack_gather_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "casmvs_pack_gather_batch_f32": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "casmvs_abn_train_finish_f32": (c_int, [_FP, c_int, c_int, c_double, _FP, _FP, c_float, c_float, c_float] + [_FP] * 6 + [c_void_p]),
    "casmvs_abn_backward_finish_f32": (c_int, [_FP, c_int, c_int, c_double, _FP, c_float] + [_FP] * 4 + [c_void_p]),
    "casmvs_abn_train_apply_f32": (c_int, [_FP, _FP, c_int, c_double, _FP, _FP, c_float, c_float, c_float] + [_FP] * 7 + [c_int, c_int, c_size_t, c_float, c_void_p]),
    "casmvs_abn_backward_apply_fused_f32": (c_int, [_FP, _FP, _FP, _FP, c_int, c_double, _FP, c_float] + [_FP] * 6 + [c_int, c_int, c_size_t, c_float, c_void_p]),
    "casmvs_upsample2x_add_f32": (c_int, [_FP, _FP, _FP, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_upsample2x_backward_f32": (c_int, [_FP, _FP, c_int, c_int, c_int, c_int, c_void_p]),
    "casmvs_costvol_backward_workspace_bytes": (c_size_t, [c_int] * 7),
    "casmvs_costvol_var_backward_f32": (c_int, [_FP] * 5 + [c_void_p] + [c_int] * 6 + [c_void_p]),
    "casmvs_costvol_gwc_backward_f32": (c_int, [_FP] * 5 + [c_void_p] + [c_int] * 7 + [c_void_p]),
    "casmvs_normalize_images_u8": (c_int, [_FP, _FP, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p]),
    "casmvs_selftest_mfma": (c_int, [_FP]),
    "casmvs_selftest_mfma_rate": (c_int, [c_int, c_int, c_int, POINTER(c_float)]),
}

# exported by -DCASMVS_TRACE builds only (tools/build_trace_lib.sh; select the build with CASMVS_LIB_PATH): bound when present
TRACE_SYMBOLS = {
    "casmvs_debug_disturb": (c_int, [c_int, c_int, c_int, c_int, _FP, c_void_p]),
}

_lib = None


class CasMVSLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise CasMVSLibraryError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m casmvsnet_pl_amd.build). "
            "casmvsnet_pl_amd has no CPU/PyTorch fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    for name, (restype, argtypes) in TRACE_SYMBOLS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = restype, argtypes
    got = lib.casmvs_abi_version()
    if got != ABI_VERSION:
        raise CasMVSLibraryError(f"libcasmvs_hip.so ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().casmvs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
