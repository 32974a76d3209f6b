"""ctypes binding of libdfmir_hip.so (the C ABI declared in include/dfmir_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel launch fails the
call raises.  Nothing here imports `oracle/`.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_void_p, POINTER, Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFMIR_HIP_LIB: an alternative build of the same C ABI (A/B experiments); default = the in-tree library
LIB_PATH = os.environ.get("DFMIR_HIP_LIB") or os.path.join(_HERE, "libdfmir_hip.so")


class DfConvGeom(Structure):
    """Mirror of `struct DfConvGeom` (include/dfmir_hip.h)."""
    _fields_ = [(n, c_int) for n in (
        "N", "Cin", "Cout", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "KD", "KH", "KW",
        "stride", "dil", "pd", "ph", "pw", "pad_mode", "act")] + [("slope", c_float)]


P = c_void_p
_GP = POINTER(DfConvGeom)
_SIGS = {
    "dfmir_abi_version": [],
    "dfmir_set_option": [ctypes.c_char_p, ctypes.c_char_p],
    "dfmir_get_option": [ctypes.c_char_p, ctypes.c_char_p, c_int],
    "dfmir_conv_fwd": [_GP, P, P, P, P, P],
    "dfmir_conv_wgrad": [_GP, P, P, P, P],
    "dfmir_bias_grad": [P, P, c_int, c_int, c_longlong, P],
    "dfmir_weight_pack": [P, P, c_int, c_int, c_int, c_int, P],
    "dfmir_weight_pack_floats": [c_int, c_int, c_int],
    "dfmir_weight_pack_batch": [P, c_int, P, c_int, P],
    "dfmir_conv3x3_res_ok": [_GP],
    "dfmir_conv3x3_fwd_scaled_res": [_GP, P, P, c_int, P, P, P, P, c_int, P, P],
    "dfmir_conv3x3_reflect_ring_ok": [_GP],
    "dfmir_conv3x3_reflect_ring_len": [_GP],
    "dfmir_conv3x3_reflect_ring": [_GP, P, P, P, c_int, P, P, P],
    "dfmir_conv_fwd_scaled": [_GP, P, P, c_int, P, P, P, P],
    "dfmir_conv_wgrad_scaled": [_GP, P, P, c_int, P, P, c_int, P, P, P],
    "dfmir_conv_wgrad_scaled_ch": [_GP, P, P, c_int, P, P, c_int, P, P, P, P],
    "dfmir_instnorm_bwd_pmax_ok": [c_longlong],
    "dfmir_instnorm_bwd_pmax": [P, P, P, P, P, c_int, c_longlong, c_int, P, P, c_int, P, P],
    "dfmir_patch_gather_bwd_gp": [P, P, P, c_int, c_int, c_longlong, c_int, c_int, P, P, P],
    "dfmir_patch_gather_bwd_any": [P, P, P, c_int, c_int, c_longlong, c_int, c_int, P, P, P],
    "dfmir_absmax": [P, c_longlong, P, P],
    "dfmir_conv3d_split_ok": [_GP],
    "dfmir_conv3d_split_ws_floats": [c_int, c_int],
    "dfmir_conv3d_split_fwd": [_GP, P, P, c_int, P, P, P, P, P, P],
    "dfmir_conv3d_split_fwd_sub": [_GP, P, P, c_int, P, P, P, P, P, c_int, P],
    "dfmir_conv3d_split_fwd_actgrad": [_GP, P, P, c_int, P, P, P, P, P, c_int, P, c_float, P],
    "dfmir_conv3d_up_ok": [c_int] * 6,
    "dfmir_conv3d_wsplit_batch": [P, c_int, P],
    "dfmir_conv3d_split_is_pair": [c_int],
    "dfmir_conv3d_march_ok": [_GP],
    "dfmir_conv3d_march_fwd": [_GP, P, P, c_int, P, P, P, P, P, c_float, P],
    "dfmir_conv3d_tiny_ok": [_GP],
    "dfmir_conv3d_tiny_fwd": [_GP, P, P, P, P, P, P, c_float, P],
    "dfmir_conv3d_s2c2_ok": [_GP],
    "dfmir_conv3d_s2c2_fwd": [_GP, P, P, P, P, P, P],
    "dfmir_conv3d_s2c2_wgrad": [_GP, P, P, P, P],
    "dfmir_conv3d_s2_ok": [_GP],
    "dfmir_conv3d_s2_fwd": [_GP, P, P, P, P, P, P],
    "dfmir_conv3d_s2_wgrad": [_GP, P, P, P, P, P],
    "dfmir_conv3d_s2_dgrad_ok": [_GP],
    "dfmir_conv3d_s2_dgrad": [_GP, P, P, P, P],
    "dfmir_conv3d_up_ws_floats": [c_int, c_int],
    "dfmir_conv3d_up_fwd": [P, P, c_int, P, c_int, P, P] + [c_int] * 6 + [P],
    "dfmir_conv3d_up_skip2_fwd": [P, P, c_int, P, P, c_int, c_int, P, P, P, P, P] + [c_int] * 7 + [c_float, P],
    "dfmir_conv3d_up_dgrad_ws_floats": [c_int, c_int],
    "dfmir_conv3d_up_dgrad": [P, P, c_int, P, c_int, P, P, P, P, c_float] + [c_int] * 6 + [P],
    "dfmir_conv3d_split_fwd_add": [_GP, P, P, c_int, P, c_int, c_int, P, P, P, P, P],
    "dfmir_conv3d_split_wgrad_ok": [_GP],
    "dfmir_conv3d_split_wgrad": [_GP, P, P, c_int, P, P, c_int, P, P],
    "dfmir_conv3d_split_wgrad_upcat": [_GP, P, P, c_int, P, c_int, P, P, c_int, P, P, P],
    "dfmir_conv3d_upwgrad_ok": [_GP, c_int],
    "dfmir_conv3d_wgrad_is_march": [_GP],
    "dfmir_conv3d_wgrad_is_march_at": [_GP, P, P],
    "dfmir_conv3d_upwgrad_ws_floats": [],
    "dfmir_conv3d_upwgrad": [_GP, P, P, c_int, P, c_int, P, P, c_int, P, P, P, P],
    "dfmir_conv3d_split_wgrad_db": [_GP, P, P, c_int, P, P, c_int, P, P, P],
    "dfmir_probe_merge": [P, P, P, P],
    "dfmir_act_bwd_amax": [P, P, P, c_longlong, c_int, c_float, P, P],
    "dfmir_weight_unpack": [P, P, c_int, c_int, c_int, P],
    "dfmir_weight_unpack_add_batch": [P, c_int, c_longlong, P],
    "dfmir_conv7x7_c1_fwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "dfmir_conv7x7_c1_wgrad": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "dfmir_tapstack_fwd": [P, P] + [c_int] * 7 + [P],
    "dfmir_tapstack_bwd": [P, P] + [c_int] * 7 + [P],
    "dfmir_tapsum_fwd": [P, P, P] + [c_int] * 10 + [c_float, P],
    "dfmir_tapsum_bwd": [P, P] + [c_int] * 9 + [P],
    "dfmir_instnorm_fwd": [P, P, P, P, P, c_int, c_longlong, c_float, c_int, P, P],
    "dfmir_instnorm_stats_ok": [c_longlong],
    "dfmir_instnorm_stats": [P, P, P, c_int, c_longlong, c_float, c_int, P, P],
    "dfmir_instnorm_bwd": [P, P, P, P, P, c_int, c_longlong, c_int, P, P],
    "dfmir_instnorm_bwd_cols_ok": [c_longlong, c_int],
    "dfmir_instnorm_bwd_cols": [P, P, P, P, P, c_int, c_longlong, c_int, P, P, c_int, P],
    "dfmir_in_relu_blurdown_ok": [c_int, c_int],
    "dfmir_in_relu_blurdown_fwd": [P, P, P, P, c_int, c_int, c_int, c_float, P, P],
    "dfmir_in_relu_blurdown_bwd": [P, P, P, P, P, c_int, c_int, c_int, P, P, P],
    "dfmir_act_bwd": [P, P, P, c_longlong, c_int, c_float, P],
    "dfmir_blur_down_fwd": [P, P, c_int, c_int, c_int, P],
    "dfmir_blur_down_bwd": [P, P, c_int, c_int, c_int, P],
    "dfmir_blur_up_fwd": [P, P, c_int, c_int, c_int, P],
    "dfmir_blur_up_bwd": [P, P, c_int, c_int, c_int, P],
    "dfmir_reflect_pad2d_fwd": [P, P, c_int, c_int, c_int, c_int, P],
    "dfmir_reflect_pad2d_bwd": [P, P, c_int, c_int, c_int, c_int, P],
    "dfmir_reflect_pad2d_bwd_add": [P, P, P, c_int, c_int, c_int, c_int, P],
    "dfmir_upcat_fwd": [P, P, P] + [c_int] * 7 + [P],
    "dfmir_upcat_bwd": [P, P, P] + [c_int] * 7 + [P],
    "dfmir_cat_channels_fwd": [P, P, P, c_longlong, c_longlong, c_longlong, P],
    "dfmir_cat_channels_bwd": [P, P, P, c_longlong, c_longlong, c_longlong, P],
    "dfmir_scale": [P, P, c_longlong, c_float, P],
    "dfmir_warp2d_fwd": [P, P, P] + [c_int] * 6 + [P],
    "dfmir_warp2d_bwd": [P, P, P, P, P] + [c_int] * 6 + [P],
    "dfmir_warp3d_fwd": [P, P, P] + [c_int] * 7 + [P],
    "dfmir_warp3d_bwd": [P, P, P, P, P] + [c_int] * 7 + [P],
    "dfmir_warp_bwd_own_ws_floats": [c_int] * 6,
    "dfmir_warp_bwd_own": [c_int, P, P, P, P, P] + [c_int] * 7 + [P, P],
    "dfmir_resize_fwd": [P, P] + [c_int] * 7 + [c_float, P],
    "dfmir_resize_bwd": [P, P] + [c_int] * 7 + [c_float, P],
    "dfmir_resize_bwd_ws_floats": [c_int] * 7,
    "dfmir_resize_bwd_sep": [P, P] + [c_int] * 7 + [c_float, P, P],
    "dfmir_patch_gather_fwd": [P, P, P, c_int, c_int, c_longlong, c_int, P],
    "dfmir_patch_gather_bwd": [P, P, P, c_int, c_int, c_longlong, c_int, P],
    "dfmir_patch_gather_bwd_amax": [P, P, P, c_int, c_int, c_longlong, c_int, P, P],
    "dfmir_patch_gather_fwd_g": [P, P, P, c_int, c_int, c_longlong, c_int, c_int, P],
    "dfmir_patch_gather_bwd_g": [P, P, P, c_int, c_int, c_longlong, c_int, c_int, P, P],
    "dfmir_l2norm_fwd": [P, P, P, c_int, c_longlong, c_float, P],
    "dfmir_l2norm_bwd": [P, P, P, P, c_int, c_longlong, c_float, P],
    "dfmir_patchnce_fwd": [P, P, P, P, c_longlong, c_int, c_int, c_float, P],
    "dfmir_patchnce_bwd": [P, P, P, P, c_longlong, c_int, c_int, c_float, P],
    "dfmir_masked_l1_fwd": [P, P, P, c_float, P, P, c_longlong, P],
    "dfmir_masked_l1_bwd": [P, P, P, c_float, P, P, P, P, c_longlong, P],
    "dfmir_flow_smooth_ws_floats": [],
    "dfmir_flow_smooth_fwd": [P, P, P] + [c_int] * 5 + [P],
    "dfmir_flow_smooth_bwd": [P, P, P] + [c_int] * 5 + [P],
    "dfmir_flow_smooth_fwd_p": [P, P, P] + [c_int] * 6 + [P],
    "dfmir_flow_smooth_bwd_p": [P, P, P] + [c_int] * 6 + [P],
    "dfmir_mul": [P, P, P, c_longlong, P],
    "dfmir_ncc_fwd_m": [P, P, P, c_int, P, P, P, P] + [c_int] * 5 + [c_float, P],
    "dfmir_ncc_bwd_m": [P, P, P, c_int, P, P, P, P, P, P] + [c_int] * 5 + [c_float, P],
    "dfmir_ncc_fwd": [P, P, P, P, P, P] + [c_int] * 5 + [c_float, P],
    "dfmir_ncc_bwd": [P, P, P, P, P, P, P, P] + [c_int] * 5 + [c_float, P],
    "dfmir_patch_gather_fwd_multi": [P, c_int, P, P, c_int, c_int, c_longlong, c_int, P],
    "dfmir_nce_head_fwd": [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_longlong, c_int, c_float, P],
    "dfmir_patch_ids_draw": [P, P, c_int, c_int, c_int, P, P],
    "dfmir_segment_means_fwd": [P, P, c_int, c_int, c_longlong, c_float, P],
    "dfmir_segment_means_bwd": [P, P, c_int, c_int, c_longlong, c_float, P],
    "dfmir_scalar_combine_fwd": [P, c_int, P, c_int, P, P],
    "dfmir_scalar_combine_bwd": [P, c_int, P, c_int, P, P],
    "dfmir_det_head_floats": [],
    "dfmir_det_begin": [P, c_longlong, P, c_longlong, P, c_int, P, c_longlong, P, c_int, ctypes.c_double, P],
    "dfmir_det_end": [P, P, c_longlong, P, c_longlong, P],
    "dfmir_fill_zero": [P, c_longlong, P],
    "dfmir_sum_scaled": [P, P, c_longlong, c_float, P],
    "dfmir_fill_from_scalar": [P, P, c_longlong, c_float, P],
    "dfmir_adam_step": [P, P, P, P, c_longlong] + [c_float] * 7 + [P],
}

_lib = None


class DfmirHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm first: it brings its own copy of the HIP runtime (torch/lib/libamdhip64.so).  Loaded AFTER this
        # library (which names /opt/rocm's) the process would hold two runtimes and the kernels here would be launched
        # on one that has enumerated no device ("no ROCm-capable device is detected" at the first launch).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise DfmirHipError(
                "libdfmir_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C dfmir_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = c_int
        h.dfmir_weight_pack_floats.restype = c_longlong
        h.dfmir_conv3d_split_ws_floats.restype = c_longlong
        h.dfmir_conv3d_up_ws_floats.restype = c_longlong
        h.dfmir_conv3d_up_dgrad_ws_floats.restype = c_longlong
        h.dfmir_conv3d_upwgrad_ws_floats.restype = c_longlong
        h.dfmir_flow_smooth_ws_floats.restype = c_longlong
        h.dfmir_warp_bwd_own_ws_floats.restype = c_longlong
        h.dfmir_resize_bwd_ws_floats.restype = c_longlong
        h.dfmir_last_error.argtypes = []
        h.dfmir_last_error.restype = ctypes.c_char_p
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().dfmir_last_error()
        raise DfmirHipError(msg.decode() if msg else "dfmir_hip error %d" % rc)


def set_option(name, value=None):
    """dfmir_set_option: a PROCESS-GLOBAL kernel-selection switch (include/dfmir_hip.h, "Options"); value None = unset."""
    check(lib().dfmir_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name):
    buf = ctypes.create_string_buffer(256)
    n = lib().dfmir_get_option(name.encode(), buf, 256)
    return None if n < 0 else buf.value.decode()


def exported_symbols():
    return sorted(list(_SIGS.keys()) + ["dfmir_last_error"])
