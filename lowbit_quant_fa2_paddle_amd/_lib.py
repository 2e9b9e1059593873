"""ctypes binding of liblowbit_fa_hip.so (C ABI declared in include/lowbit_fa.h).

The library is the product; there is no CPU or eager fallback.  If it cannot be loaded every
operator raises `RuntimeError` telling the user how to build it.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "liblowbit_fa_hip.so"
# LBFA_LIB_PATH: development override to A/B kernel builds (same C ABI)
LIB_PATH = os.environ.get("LBFA_LIB_PATH") or os.path.join(_HERE, LIB_NAME)

LBFA_F16, LBFA_BF16, LBFA_E4M3 = 0, 1, 2
LBFA_OK, LBFA_EINVAL, LBFA_ELAUNCH = 0, 1, 2
BLKQ, BLKK = 128, 64

_i64x3 = ctypes.c_int64 * 3
_i64x2 = ctypes.c_int64 * 2
_vp, _ci, _cf, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_cd = ctypes.c_double

# name -> (restype, argtypes); must list every symbol declared in include/lowbit_fa.h
SIGNATURES = {
    "lbfa_version": (_ci, []),
    "lbfa_last_error": (ctypes.c_char_p, []),
    "lbfa_mean_seq_workspace_bytes": (_sz, [_ci, _ci, _ci, _ci]),
    "lbfa_mean_seq": (_ci, [_vp, _ci, _vp, _vp, _sz, _ci, _ci, _ci, _ci, ctypes.POINTER(ctypes.c_int64), _vp]),
    "lbfa_quant_per_block": (_ci, [_vp, _ci, _vp, _ci, _vp, _vp, _cf, _ci, _ci, _ci, _ci, _ci, _ci,
                                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                   _vp, _ci, _vp, _vp]),
    "lbfa_v_fp8_bytes": (_sz, [_ci, _ci, _ci, _ci]),
    "lbfa_quant_v_fp8": (_ci, [_vp, _ci, _vp, _vp, _ci, _ci, _ci, _ci, ctypes.POINTER(ctypes.c_int64), _vp]),
    "lbfa_attn_fwd": (_ci, [_vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci,
                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _ci, _vp]),
    "lbfa_profile_next_attn": (_ci, [_vp, _vp]),
    "lbfa_absmax": (_ci, [_vp, _ci, _vp, _ci, _ci, _ci, _ci, ctypes.POINTER(ctypes.c_int64), _vp]),
    "lbfa_cast_bf16_to_f16": (_ci, [_vp, _vp, _ci, _ci, _ci, _ci, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _vp]),
    "lbfa_forward_workspace_bytes": (_sz, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci]),
    "lbfa_forward_workspace_bytes_dt": (_sz, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci]),
    "lbfa_forward": (_ci, [_vp, _vp, _vp, _ci, _vp, _vp, _vp, _sz, _ci, _ci, _ci, _ci, _ci, _ci,
                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                           _cd, _ci, _ci, _ci, _ci, _ci, _vp]),
    "lbfa_sdpa_fwd": (_ci, [_vp, _vp, _vp, _ci, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci,
                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _cd, _ci, _vp]),
    "lbfa_quant_per_block_varlen": (_ci, [_vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _cf, _ci, _ci, _ci, _ci, _ci, _ci,
                                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _vp]),
    "lbfa_attn_fwd_varlen": (_ci, [_vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _ci, _ci, _ci, _ci, _ci, _ci,
                                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _ci, _vp]),
    "lbfa_forward_varlen_workspace_bytes": (_sz, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci]),
    "lbfa_forward_varlen_workspace_bytes_dt": (_sz, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci]),
    "lbfa_forward_varlen": (_ci, [_vp, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _sz,
                                  _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci,
                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                  _cd, _ci, _ci, _ci, _ci, _vp]),
}

_lock = threading.Lock()
_lib = None


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError if the HIP library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_NAME} not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {os.path.join(_HERE, 'csrc')}` (needs hipcc, --offload-arch=gfx950). "
                "There is no CPU fallback for this operator.")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise RuntimeError(f"failed to load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def strides3(s):
    return _i64x3(int(s[0]), int(s[1]), int(s[2]))


def strides2(s):
    return _i64x2(int(s[0]), int(s[1]))


def check(status: int, lib=None):
    """Map a C-ABI status to the exception the reference would raise (ValueError for bad arguments,
    as src/core.py:287; RuntimeError for launch failures, as TORCH_CHECK -> RuntimeError)."""
    if status == LBFA_OK:
        return
    lib = lib or load()
    msg = lib.lbfa_last_error().decode("utf-8", "replace")
    if status == LBFA_EINVAL:
        raise ValueError(msg)
    raise RuntimeError(msg)
