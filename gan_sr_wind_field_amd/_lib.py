"""ctypes binding of the C-ABI in ``include/windsr_hip.h`` (libwindsr_hip.so).

There is deliberately NO fallback: if the gfx950 library is missing or a call
fails, a ``RuntimeError`` is raised.  The product path never routes through the
CPU oracle or through ATen/MIOpen convolutions.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WSR_LIB_PATH") or os.path.join(_HERE, "csrc", "libwindsr_hip.so")  # (override: tuning builds)

WSR_F32, WSR_BF16 = 0, 1
WSR_EUNSUPPORTED = -2

EXPORTS = [
    "wsr_abi_version", "wsr_error_string", "wsr_reload_env", "wsr_ragan_loss", "wsr_bn_train_stats", "wsr_conv3d_fwd", "wsr_conv3d_dgrad", "wsr_conv3d_wgrad", "wsr_conv3d_wgrad_tri", "wsr_conv3d_wgrad_nparts", "wsr_conv3d_wgrad_parts", "wsr_conv3d_wgrad_parts_x2", "wsr_conv_split_ok", "wsr_unpack_wgrad_reduce_multi", "wsr_conv3d_fwd_tile", "wsr_conv3d_dgrad_tile",
    "wsr_frag_filter_elems", "wsr_pack_filter_frag", "wsr_pack_filter_frag_multi",
    "wsr_pack_filter", "wsr_unpack_wgrad", "wsr_unpack_wgrad_multi", "wsr_lrelu_bwd_inplace", "wsr_chan_axpby", "wsr_chan_sum", "wsr_chan_sum_rows", "wsr_chan_sum_partials", "wsr_upsample2_bwd", "wsr_subpixel_fold", "wsr_subpixel_unfold", "wsr_strided_parity_filters", "wsr_strided_parity_unfold",
    "wsr_planar_to_ndhwc", "wsr_ndhwc_to_planar", "wsr_zfold", "wsr_zunfold", "wsr_wind_gradient", "wsr_wind_gradient_bwd", "wsr_plane_sum", "wsr_linear_rows", "wsr_physics_loss_workspace_floats", "wsr_physics_loss_stats", "wsr_physics_loss_bwd", "wsr_bn_stats", "wsr_bn_mean", "wsr_bn_shard_stats", "wsr_bn_combine_shards", "wsr_bn_finalize", "wsr_bn_apply_lrelu", "wsr_bn_bwd_reduce",
    "wsr_bn_bwd_apply", "wsr_adam_step", "wsr_adam_multi",
]


class ConvDesc(C.Structure):
    """``wsr_conv_t``."""

    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "B", "Xi", "Yi", "Zi", "Xo", "Yo", "Zo", "Cin", "in_ctot", "in_off", "Cout", "out_ctot",
        "out_off", "KX", "KY", "KZ", "sx", "sy", "sz", "px", "py", "pz", "upsample_xy", "lat", "lat_ox", "lat_oy",
        "lat_phases", "lat_mz", "lat_oz")]


class Epilogue(C.Structure):
    """``wsr_epilogue_t``."""

    _fields_ = [
        ("bias", C.c_void_p), ("chan_scale", C.c_void_p), ("res", C.c_void_p),
        ("res_ctot", C.c_int32), ("res_off", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float),
        ("act", C.c_int32), ("slope", C.c_float), ("out_planar", C.c_int32), ("act_c1", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("res2", C.c_void_p), ("res2_ctot", C.c_int32), ("res2_off", C.c_int32), ("beta2", C.c_float),
        ("mask", C.c_void_p),
        ("in2", C.c_void_p), ("in2_ctot", C.c_int32), ("in2_c0", C.c_int32),   # ABI 8: the concat as two tensors
    ]


class LreluMask(C.Structure):
    """``wsr_lrelu_mask_t``."""

    _fields_ = [("y", C.c_void_p), ("y_ctot", C.c_int32), ("y_off", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
                ("slope", C.c_float), ("chan_scale", C.c_void_p)]


class DgradOpts(C.Structure):
    """``wsr_dgrad_opts_t`` (ABI 6)."""

    _fields_ = [("acc_src", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("acc_beta", C.c_float),
                ("beta2", C.c_float), ("res2", C.c_void_p), ("res2_ctot", C.c_int32), ("res2_off", C.c_int32),
                ("dx2", C.c_void_p), ("dx2_ctot", C.c_int32), ("dx2_c0", C.c_int32)]   # ABI 8


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the gfx950 kernels first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C gan_sr_wind_field_amd/csrc). "
            "There is no CPU / ATen fallback for the hot path.")
    # PyTorch's own HIP runtime must be in the process BEFORE this library is: libwindsr_hip.so needs libamdhip64 by
    # soname, and loaded first it would bring in the system's copy - two runtimes, and the kernels launch into the one
    # that holds no device ("no ROCm-capable device", seen with build() and smoke() in one process)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.wsr_abi_version.restype = C.c_int
    L.wsr_error_string.restype = C.c_char_p
    L.wsr_error_string.argtypes = [C.c_int]
    sig = {
        "wsr_conv3d_fwd": [C.POINTER(ConvDesc), vp, vp, vp, C.POINTER(Epilogue), vp],
        "wsr_conv3d_dgrad": [C.POINTER(ConvDesc), vp, vp, vp, f32, C.c_int, C.c_int, vp],
        "wsr_conv3d_wgrad": [C.POINTER(ConvDesc), vp, vp, vp, vp],
        "wsr_conv3d_wgrad_tri": [C.POINTER(ConvDesc), vp, vp, vp, i32, i32, vp],
        "wsr_conv3d_wgrad_nparts": [C.POINTER(ConvDesc), i32, i32, C.POINTER(C.c_int32)],
        "wsr_conv3d_wgrad_parts": [C.POINTER(ConvDesc), vp, vp, vp, i64, i32, i32, i32, vp],
        "wsr_conv3d_wgrad_parts_x2": [C.POINTER(ConvDesc), vp, vp, i32, i32, vp, vp, i64, i32, vp],
        "wsr_conv_split_ok": [C.POINTER(ConvDesc), i32],
        "wsr_unpack_wgrad_reduce_multi": [vp, i32, vp],
        "wsr_conv3d_fwd_tile": [C.POINTER(ConvDesc), vp, vp, vp, C.POINTER(Epilogue), vp],
        "wsr_conv3d_dgrad_tile": [C.POINTER(ConvDesc), vp, vp, vp, f32, C.c_int, C.c_int, C.POINTER(LreluMask),
                                  C.POINTER(DgradOpts), vp],
        "wsr_pack_filter_frag": [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
        "wsr_pack_filter_frag_multi": [vp, i32, i32, vp],
        "wsr_pack_filter": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
        "wsr_unpack_wgrad": [vp, vp, i32, i32, i32, i32, f32, i32, vp],
        "wsr_unpack_wgrad_multi": [vp, i32, vp],
        "wsr_lrelu_bwd_inplace": [vp, i32, i32, vp, i32, i32, i32, i64, f32, vp, i64, i32, vp],
        "wsr_chan_axpby": [vp, i32, i32, vp, i32, i32, i32, i64, f32, f32, i32, vp],
        "wsr_chan_sum": [vp, i32, i32, i32, i64, f32, vp, vp, i32, vp],
        "wsr_chan_sum_rows": [i32, i64],
        "wsr_chan_sum_partials": [vp, i32, i32, i32, i64, vp, i32, vp],
        "wsr_upsample2_bwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
        "wsr_subpixel_fold": [vp, vp, i64, i32, vp],
        "wsr_strided_parity_filters": [vp, vp, i32, i32, i32, i32, vp],
        "wsr_subpixel_unfold": [vp, vp, i64, i32, vp],
        "wsr_planar_to_ndhwc": [vp, vp, i32, i32, i64, i32, i32, i32, i32, vp],
        "wsr_ndhwc_to_planar": [vp, vp, i32, i32, i64, i32, i32, i32, vp],
        "wsr_zfold": [vp, vp, vp, i32, i32, i32, i32, i64, i32, vp],
        "wsr_wind_gradient": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "wsr_wind_gradient_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "wsr_plane_sum": [vp, i32, i32, i64, vp, vp, vp],
        "wsr_linear_rows": [vp, vp, vp, vp, i32, i32, i64, vp],
        "wsr_physics_loss_stats": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "wsr_physics_loss_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "wsr_zunfold": [vp, vp, i32, i32, i32, i32, i64, i32, i32, i32, i32, i32, vp],
        "wsr_bn_stats": [vp, i32, i64, vp, vp, vp, i32, vp],
        "wsr_bn_mean": [vp, vp, f32, vp, i32, vp],
        "wsr_bn_shard_stats": [vp, i32, vp, i32, f32, vp, i32, i32, vp],
        "wsr_bn_combine_shards": [vp, i32, f32, vp, i32, vp, i32, i32, i32, vp],
        "wsr_bn_finalize": [vp, vp, f32, vp, f32, f32, vp, vp, vp, vp, i32, vp],
        "wsr_bn_apply_lrelu": [vp, vp, vp, vp, vp, vp, i32, i64, i32, f32, i32, vp],
        "wsr_bn_bwd_reduce": [vp, vp, vp, vp, vp, i32, i64, i32, f32, vp, vp, i32, vp],
        "wsr_bn_bwd_apply": [vp, vp, vp, vp, vp, vp, vp, f32, vp, f32, i32, i64, i32, vp],
        "wsr_adam_step": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp],
        "wsr_adam_multi": [vp, i32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i32, vp],
        "wsr_ragan_loss": [vp, vp, vp, vp, vp, vp, i32, vp, vp],   # ABI 9
        "wsr_bn_train_stats": [vp, i32, i64, i32, f32, f32, vp, vp, vp, vp, i32, vp],   # ABI 9
    }
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    L.wsr_frag_filter_elems.argtypes = [i32, i32, i32, i32]
    L.wsr_frag_filter_elems.restype = C.c_int64
    L.wsr_physics_loss_workspace_floats.argtypes = []
    L.wsr_physics_loss_workspace_floats.restype = C.c_int64
    if L.wsr_abi_version() != 9:
        raise RuntimeError("libwindsr_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().wsr_error_string(rc).decode()
        raise RuntimeError(f"windsr_hip {what} failed: {msg} (code {rc})")
