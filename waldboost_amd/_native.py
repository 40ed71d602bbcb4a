"""ctypes binding of the C ABI declared in include/waldboost_hip.h.

There is no CPU fallback anywhere in this package: if the HIP library has not been
built (``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C
waldboost_amd/csrc``) every compute entry point raises ``RuntimeError``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WB_NATIVE_LIB: A/B timing of two builds of the same ABI in one session (diagnostic)
LIB_PATH = os.environ.get("WB_NATIVE_LIB") or os.path.join(_HERE, "csrc", "libwaldboost_hip.so")

WB_DTYPE_U8, WB_DTYPE_F32, WB_DTYPE_RANK8 = 0, 1, 2
WB_DTYPE_RANK16 = 13
WB_DTYPE_F64, WB_DTYPE_I8, WB_DTYPE_I16, WB_DTYPE_U16, WB_DTYPE_I32, WB_DTYPE_U32 = 3, 4, 5, 6, 7, 8
WB_ERR_INVALID, WB_ERR_HIP, WB_ERR_UNSUPPORTED = -1, -2, -3
WB_DET_SHARDS = 64
WB_DTYPE_I64, WB_DTYPE_U64, WB_DTYPE_BOOL, WB_DTYPE_F16 = 9, 10, 11, 12
WB_CHN_GRAD_HIST, WB_CHN_GRAD_HIST_4_U1, WB_CHN_GRAD_MAG_U1, WB_CHN_GRAD_MAG = 0, 1, 2, 3
WB_ABI_VERSION = 8

# numpy mirrors of the ABI structs (sizes asserted against the header's comments)
LEVEL_DTYPE = np.dtype([
    ("oct", "<i4"), ("src_h", "<i4"), ("src_w", "<i4"), ("nh", "<i4"), ("nw", "<i4"),
    ("u", "<i4"), ("v", "<i4"), ("tap_off", "<i4"), ("src_off", "<i8"), ("chn_off", "<i8"),
    ("sy", "<f8"), ("sx", "<f8")], align=True)
TAP_DTYPE = np.dtype([("i0", "<i4"), ("i1", "<i4"), ("w0", "<f8"), ("w1", "<f8")], align=True)
TILE_DTYPE = np.dtype([("level", "<i4"), ("ty", "<u2"), ("tx", "<u2")], align=True)
PATCH_DTYPE = np.dtype([("r_lo", "<i4"), ("c_lo", "<i4"), ("rows", "<u2"), ("bytes", "<u2"), ("pad", "<u4")], align=True)   # WbTilePatch
DET_DTYPE = np.dtype([("image", "<i4"), ("level", "<i4"), ("r", "<u2"), ("c", "<u2"), ("score", "<f4")], align=True)
assert LEVEL_DTYPE.itemsize == 64 and TILE_DTYPE.itemsize == 8 and DET_DTYPE.itemsize == 16 and TAP_DTYPE.itemsize == 24
assert PATCH_DTYPE.itemsize == 16


class WbModelInfo(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("depth", C.c_int32), ("m", C.c_int32), ("n", C.c_int32),
                ("C", C.c_int32), ("tile_rows", C.c_int32), ("tile_cols", C.c_int32), ("lds_bytes", C.c_int32),
                ("rank_ok", C.c_int32), ("specialized", C.c_int32), ("rank16_ok", C.c_int32)]


# every symbol include/waldboost_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "wb_abi_version": (C.c_int, []),
    "wb_last_error": (C.c_char_p, []),
    "wb_channels_tile": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "wb_channel_func_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "wb_octaves_launch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, C.c_int64,
                                    C.POINTER(C.c_int64), C.c_int, _P]),
    "wb_octaves_launch_z": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, C.c_int64,
                                      C.POINTER(C.c_int64), C.c_int, _P, _P, C.c_int]),
    "wb_channels_launch": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int,
                                     _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _P, C.c_int64,
                                     _P, _P, C.c_int64]),
    "wb_channels_launch_x": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int,
                                       _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _P, C.c_int64,
                                       _P, _P, C.c_int64, _P, C.c_int]),
    "wb_channels_tile_patches": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P]),
    "wb_resize_level_launch": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "wb_pool_smooth_launch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "wb_grad_hist_launch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), _P]),
    "wb_grad_mag_launch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_double, C.c_int, _P, _P]),
    "wb_model_create": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "wb_model_destroy": (C.c_int, [_P]),
    "wb_model_info": (C.c_int, [_P, C.POINTER(WbModelInfo)]),
    "wb_model_specialize": (C.c_int, [_P, C.c_int]),
    "wb_model_use_specialized": (C.c_int, [_P, C.c_int]),
    "wb_jit_compile_check": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int64)]),
    "wb_jit_compile_check2": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int64)]),
    "wb_rankgroup_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(_P)]),
    "wb_rankgroup_model": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "wb_rankgroup_destroy": (C.c_int, [_P]),
    "wb_cascade_launch": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P,
                                    C.c_uint32, _P]),
    "wb_cascade_launch_z": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P,
                                      C.c_uint32, _P, _P, C.c_int]),
    "wb_tree_eval_launch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, _P, _P, _P, _P, _P,
                                      C.c_int, _P]),
    "wb_gather_samples_launch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "wb_samples_predict_launch": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P]),
    "wb_tree_apply_launch": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P]),
    "wb_boxes_launch": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int, C.c_int, _P, _P]),
    "wb_det_pack_launch": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32]),
    "wb_det_finish_launch": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_uint32]),
    "wb_det_finish_sorted_launch": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_uint32, _P,
                                              C.c_uint32]),
    "wb_det_order_batch_launch": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t,
                                            _P, C.c_uint32]),
    "wb_selftest_projection": (C.c_int, [_P, _P]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def load():
    """Load libwaldboost_hip.so (once).  torch is imported first so that the library binds to
    the HIP runtime already in the process (same soname) instead of a second copy."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `make -C waldboost_amd/csrc` or __graft_entry__.build()). "
            "waldboost_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64.so.7 that our library resolves against)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.wb_abi_version() != WB_ABI_VERSION:
        raise NativeError("libwaldboost_hip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error():
    """The thread's last error message of the native library (wb_last_error)."""
    return load().wb_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc == 0:
        return
    msg = load().wb_last_error().decode("utf-8", "replace")
    if rc == WB_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if rc == WB_ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    raise NativeError(f"{what}: {msg} (code {rc})")


def require_gpu():
    """The torch device this process computes on; raises when there is no GPU."""
    import torch
    if not torch.cuda.is_available():
        raise NativeError("waldboost_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                          "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
