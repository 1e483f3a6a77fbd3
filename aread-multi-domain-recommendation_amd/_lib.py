"""ctypes binding of libaread_hip.so (include/aread_hip.h).

There is no CPU fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.  Status codes are turned into RuntimeError(aread_last_error()), the exception class the
reference's callers already see from PyTorch (error conventions)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AREAD_HIP_LIB") or os.path.join(_HERE, "libaread_hip.so")   # env override: A/B builds
_lib = None

i32p, f32p, vp = C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers


class PlanLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("max_rows", "max_tiles", "words", "off_seg_count", "off_seg_start",
                                         "off_tile_seg", "off_tile_valid", "off_row_sample", "off_sample_row")]


_SIGS = {
    "aread_version": (C.c_int, []),
    "aread_last_error": (C.c_char_p, []),
    "aread_plan_layout_get": (C.c_int, [C.c_int64, C.c_int, C.POINTER(PlanLayout)]),
    "aread_plan_build": (C.c_int, [i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, vp]),
    "aread_embed_fwd": (C.c_int, [i32p, C.c_int64, C.c_int, i32p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, i32p, C.c_int64, f32p, i32p, vp]),
    "aread_embed_bwd_ws_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "aread_embed_bwd": (C.c_int, [i32p, C.c_int64, C.c_int, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, i32p, f32p, f32p, vp, vp]),
    "aread_embed_bwd_sort": (C.c_int, [i32p, C.c_int64, C.c_int, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, i32p, vp, vp]),
    "aread_embed_bwd_reduce": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int, f32p, f32p, vp, vp]),
    "aread_embed_bwd_reduce2": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, vp, vp]),
    "aread_route_ws_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "aread_route_build": (C.c_int, [i32p, C.c_int64, C.c_int, i32p, C.c_int64, C.c_int, vp, i32p, i32p, i32p, C.c_int, vp]),
    "aread_adam_step": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int64, vp, vp, vp]),
    "aread_adam_table_l2": (C.c_int, [f32p, f32p, f32p, C.c_int64, C.c_int, vp, i32p, i32p, f32p, C.c_float, vp, C.c_int, f32p,
                                      vp]),
    "aread_adam_row_partials": (C.c_int, []),
    "aread_l2_partials": (C.c_int, []),
    "aread_l2_table": (C.c_int, [f32p, C.c_int64, C.c_float, C.c_float, f32p, f32p, vp]),
    "aread_l2_table_throttled": (C.c_int, [f32p, C.c_int64, C.c_float, C.c_float, f32p, f32p, C.c_int, vp]),
    "aread_l2_table_dev": (C.c_int, [f32p, C.c_int64, C.c_float, f32p, f32p, vp]),
    "aread_l2_finish": (C.c_int, [f32p, C.c_int, C.c_float, f32p, C.c_int, vp]),
    "aread_gemm_bf16x3": (C.c_int, [f32p, C.c_int64, C.c_int64, f32p, C.c_int64, C.c_int64, f32p, C.c_int64, C.c_int64, f32p,
                                    C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "aread_gemm_bf16x3_rc": (C.c_int, [f32p, C.c_int64, C.c_int64, f32p, C.c_int64, C.c_int64, f32p, C.c_int64, C.c_int64, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "aread_debug_set": (C.c_int, [C.c_char_p, C.c_int]),
    "aread_debug_get": (C.c_longlong, [C.c_char_p]),
    "aread_debug_phase_times": (C.c_int, [C.c_void_p, C.c_int]),
    "aread_debug_gather_roof": (C.c_int, [i32p, C.c_int64, f32p, C.c_int, f32p, C.c_int64, vp]),
    "aread_gemm": (C.c_int, [f32p, C.c_int64, C.c_int64, C.c_int, f32p, C.c_int64, C.c_int64, C.c_int, f32p, C.c_int64,
                             C.c_int64, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
}


def exported_symbols():
    """Every symbol include/aread_hip.h declares (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def register(name, restype, argtypes):
    _SIGS[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               f"g.build()'` (or make -C {os.path.join(_HERE, 'csrc')}); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = restype, argtypes
    return _lib


def check(status):
    if status != 0:
        raise RuntimeError(f"libaread_hip: {lib().aread_last_error().decode()} (status {status})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("aread_amd runs on HIP devices only: got a CPU tensor (there is no CPU fallback; "
                               "move the module and its inputs to 'cuda')")


def require(t, dtype, name):
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    return t
