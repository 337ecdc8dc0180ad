"""ctypes binding of libdvc_hip.so (the C-ABI declared in include/dvc_hip.h).

There is deliberately NO CPU / eager-PyTorch fallback: if the shared library is missing or an input
is not a ROCm device tensor, the call raises.  (The CPU restatement lives in oracle/ and is test
infrastructure only.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVC_DEBUG_LIB=1 (tools/ only) loads the -DDVC_DEBUG build, the only one that carries the dvc_debug_* hooks
DEBUG_BUILD = os.environ.get("DVC_DEBUG_LIB", "0") == "1"
LIB_PATH = os.path.join(_HERE, "libdvc_hip_debug.so" if DEBUG_BUILD else "libdvc_hip.so")
ABI_VERSION = 19

c_float_p = ctypes.c_void_p
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64


class DvcConvDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("Cin", c_i32), ("H", c_i32), ("W", c_i32),
        ("Cout", c_i32), ("ksize", c_i32), ("stride", c_i32), ("dil", c_i32),
        ("pad", c_i32), ("pad_mode", c_i32), ("in_up", c_i32), ("in_sub", c_i32),
        ("act", c_i32), ("act_slope", ctypes.c_float), ("in_prelu", c_i32), ("cfg", c_i32), ("split_k", c_i32),
        ("x_batch_stride", c_i64), ("y_batch_stride", c_i64), ("res_batch_stride", c_i64),
        ("flags", c_i32), ("w_batch_stride", c_i64),
    ]


class DvcConvGroupItem(ctypes.Structure):
    _fields_ = [
        ("d", DvcConvDesc), ("x", ctypes.c_void_p), ("u_packed", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("act_slope_ptr", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("y", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
    ]


class DvcInstNormItem(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("S", c_i32), ("bias", ctypes.c_void_p), ("act", c_i32), ("act_slope", ctypes.c_float),
        ("act_slope_ptr", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("slope_ptr", ctypes.c_void_p),
        ("chan_scale", ctypes.c_void_p), ("eps", ctypes.c_float),
        ("N", c_i32), ("C", c_i32), ("H", c_i32), ("W", c_i32), ("up", c_i32), ("sub", c_i32), ("rpad", c_i32),
        ("x_batch_stride", c_i64), ("res_batch_stride", c_i64), ("y_batch_stride", c_i64), ("y", ctypes.c_void_p),
    ]


# every symbol include/dvc_hip.h declares: name -> (restype, argtypes)
_VP = ctypes.c_void_p
SIGNATURES = {
    "dvc_abi_version": (ctypes.c_int, []),
    "dvc_last_error": (ctypes.c_char_p, []),
    "dvc_conv2d_out_hw": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "dvc_conv2d": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                  ctypes.c_size_t, _VP]),
    "dvc_winograd_weight_floats": (ctypes.c_size_t, [c_i32, c_i32]),
    "dvc_winograd_pack_weight": (ctypes.c_int, [_VP, c_i32, c_i32, _VP, _VP]),
    "dvc_conv2d_winograd_split": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc), ctypes.c_size_t, ctypes.POINTER(c_i32),
                                                ctypes.POINTER(c_i32)]),
    "dvc_conv2d_winograd": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                           ctypes.c_size_t, _VP]),
    "dvc_conv2d_winograd_pool": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, c_i64, _VP,
                                                ctypes.c_size_t, _VP]),
    "dvc_conv2d_winograd_dual": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc), ctypes.POINTER(DvcConvDesc), _VP, _VP, _VP, _VP, _VP,
                                                _VP, _VP, ctypes.c_size_t, _VP]),
    "dvc_conv2d_ws_eligible": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc)]),
    "dvc_conv2d_ws_pack_weight": (ctypes.c_int, [_VP, c_i32, c_i32, _VP, _VP]),
    "dvc_conv2d_ws": (ctypes.c_int, [ctypes.POINTER(DvcConvDesc), _VP, _VP, _VP, _VP, _VP, _VP]),
    "dvc_conv2d_winograd_group": (ctypes.c_int, [ctypes.POINTER(DvcConvGroupItem), c_i32, _VP]),
    "dvc_instnorm_apply_group": (ctypes.c_int, [ctypes.POINTER(DvcInstNormItem), c_i32, _VP]),
    "dvc_conv1x1_small": (ctypes.c_int, [_VP, _VP, _VP, c_i32, c_i32, c_i32, c_i32, c_i32, _VP, _VP]),
    "dvc_instnorm_stats": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, c_i64, ctypes.c_float, _VP, _VP, _VP, _VP]),
    "dvc_affine_act": (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                      c_i64, c_i64, c_i64, _VP, _VP]),
    "dvc_instnorm_apply": (ctypes.c_int, [_VP, _VP, _VP, _VP, ctypes.c_float, c_i32, c_i32, c_i32, c_i32, c_i32,
                                          c_i32, c_i32, c_i64, c_i64, c_i64, _VP, _VP, _VP, _VP, c_i32, _VP, _VP]),
    "dvc_cx_prepare": (ctypes.c_int, [_VP, _VP, c_i32, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP, _VP, _VP]),
    "dvc_cx_rows": (ctypes.c_int, [_VP, c_i32, c_i64, c_i64, c_i32, c_i32, ctypes.c_float, _VP, _VP, _VP, _VP, _VP, _VP]),
    "dvc_cx_colmax": (ctypes.c_int, [_VP, c_i32, c_i64, c_i64, _VP, _VP, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP, _VP]),
    "dvc_cx_finish": (ctypes.c_int, [_VP, c_i32, c_i32, _VP, _VP, _VP]),
    "dvc_cx_rows_tq": (ctypes.c_int, [_VP, c_i32, c_i64, c_i64, c_i64, _VP, _VP, _VP, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP,
                                      _VP]),
    "dvc_cx_ds": (ctypes.c_int, [_VP, c_i32, c_i64, c_i64, c_i64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, ctypes.c_float, c_i32,
                                 c_i32, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP, _VP]),
    "dvc_cx_normalize_bwd": (ctypes.c_int, [_VP, _VP, _VP, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP]),
    "dvc_instnorm_apply_partials": (ctypes.c_int, [_VP, c_i32, _VP, c_i32, ctypes.c_float, _VP, _VP, _VP, _VP, ctypes.c_float,
                                                   c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, _VP, _VP, _VP,
                                                   _VP, c_i32, _VP, _VP]),
    "dvc_maxpool2x2": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, _VP, _VP]),
    "dvc_avgpool2x2": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, _VP, _VP]),
    "dvc_avgpool4x4": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, _VP, _VP]),
    "dvc_upsample_nearest": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, c_i32, _VP, _VP]),
    "dvc_channel_l2norm": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP]),
    "dvc_channel_l2norm_multi": (ctypes.c_int, [_VP, _VP, _VP, _VP, c_i32, c_i32, ctypes.c_float, _VP]),
    "dvc_gray2rgb": (ctypes.c_int, [_VP, c_i32, c_i32, c_i64, _VP, _VP]),
    "dvc_lab2rgb": (ctypes.c_int, [_VP, c_i32, c_i32, ctypes.c_float, _VP, _VP]),
    "dvc_pack_color_input": (ctypes.c_int, [_VP, c_i64, _VP, _VP, _VP, c_i64, _VP, c_i64, c_i32, c_i32, _VP, _VP]),
    "dvc_upsample_bilinear2x": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP]),
    "dvc_lum_guide_u8": (ctypes.c_int, [_VP, c_i64, _VP, _VP]),
    "dvc_fgs_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "dvc_fgs_filter": (ctypes.c_int, [_VP, _VP, c_i32, c_i32, c_i32, c_i32, ctypes.c_float, ctypes.c_float, c_i32,
                                      ctypes.c_float, _VP, _VP, ctypes.c_size_t, _VP]),
    "dvc_lab2rgb_u8": (ctypes.c_int, [_VP, _VP, c_i32, c_i32, _VP, _VP]),
    "dvc_rgb8_to_lab": (ctypes.c_int, [_VP, c_i32, c_i32, _VP, _VP]),
    "dvc_center_pad_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32]),
    "dvc_center_pad_is_fused": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32]),
    "dvc_center_pad": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, c_i32, _VP, _VP, ctypes.c_size_t, _VP]),
    "dvc_corr_prepare": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP, _VP]),
    "dvc_corr_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32]),
    "dvc_corr_fwd": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_float, ctypes.c_float, c_i32, c_i32, c_i32, c_i32,
                                    _VP, _VP, _VP, _VP, _VP, _VP, ctypes.c_size_t, _VP]),
    "dvc_corr_merge_pack": (ctypes.c_int, [_VP, ctypes.c_size_t, ctypes.c_float, c_i32, c_i32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "dvc_corr_prepare_bf16": (ctypes.c_int, [_VP, c_i32, c_i32, c_i32, ctypes.c_float, _VP, _VP, _VP, _VP]),
    "dvc_corr_bf16_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32]),
    "dvc_corr_fwd_bf16": (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, ctypes.c_float, c_i32, c_i32, c_i32, c_i32,
                                         _VP, _VP, _VP, _VP, _VP, _VP, ctypes.c_size_t, _VP]),
    "dvc_corr_softmax_bwd": (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, ctypes.c_float, ctypes.c_float, c_i32, c_i32, c_i32, c_i64, c_i32,
                                            _VP, _VP, _VP, _VP]),
}
# diagnostics for tools/ (include/dvc_hip.h, last section): exported by the -DDVC_DEBUG build only
DEBUG_SIGNATURES = {
    "dvc_debug_conv_trace": (None, [_VP]),
    "dvc_debug_conv_variant": (None, [ctypes.c_int]),
    "dvc_debug_corr_timeline": (None, [_VP, ctypes.c_int]),
    "dvc_debug_corr_variant": (None, [ctypes.c_int]),
}

_lib = None


def load():
    """Load libdvc_hip.so (once) and attach the signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C csrc`"
            f"{' DEBUG=1' if DEBUG_BUILD else ''}). "
            "There is no CPU fallback for the HIP path.")
    # torch ships its own ROCm runtime (libamdhip64 / libhsa-runtime64); it must be the one already
    # mapped when our library's DT_NEEDED entries are resolved, otherwise two HIP runtimes coexist in
    # the process and ours sees "no ROCm-capable device".
    import torch  # noqa: F401
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = ctypes.CDLL(LIB_PATH)
    table = dict(SIGNATURES)
    if DEBUG_BUILD:
        table.update(DEBUG_SIGNATURES)
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError here == symbol missing from the .so
        fn.restype = res
        fn.argtypes = args
    got = lib.dvc_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libdvc_hip.so ABI version {got} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


# DVC_SYNC_DEBUG=1: synchronise after every library call and name it on stderr first — an asynchronous GPU memory fault is
# then reported right after the call that caused it (debugging only: it serialises everything)
_SYNC_DEBUG = os.environ.get("DVC_SYNC_DEBUG", "0") == "1"


def check(rc, what=""):
    if _SYNC_DEBUG:
        import sys
        import torch
        sys.stderr.write(f"[dvc] {what}\n")
        sys.stderr.flush()
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
    if rc != 0:
        msg = load().dvc_last_error().decode(errors="replace")
        raise RuntimeError(f"libdvc_hip {what} failed: {msg}")
