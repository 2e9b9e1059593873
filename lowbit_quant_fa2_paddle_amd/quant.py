"""`per_channel_fp8` with the reference's name and return convention (src/quant.py:210-291)."""
from __future__ import annotations

from . import _lib
from ._tensor import ops_for
from .quant_per_block import _bhs


class Fp8V:
    """FP8 V in the device layout lbfa_attn_fwd consumes ([B,H,ceil(S/64),D,64] e4m3 bytes, see
    include/lowbit_fa.h).  Opaque to callers, like the reference's transposed/permuted `v_fp8`."""

    def __init__(self, buf, B, H, S, D):
        self.buf, self.B, self.H, self.S, self.D = buf, B, H, S, D


def per_channel_fp8(v, tensor_layout="HND", scale_max=448.0, smooth_v=False):
    """Per-(b,h,d) channel scale = amax/448 and e4m3 codes (csrc/fused/fused.cu:317-428).
    Returns (v_fp8, v_scale [B,H,D] fp32, vm) with vm = None: `smooth_v` is ignored exactly as the
    reference does for its default fp32+fp32 accumulation (src/core.py:879-881) - MFMA accumulates PV
    in full fp32, so the mean-subtraction work-around for NVIDIA's fp8 accumulator is not needed."""
    if scale_max != 448.0:
        raise ValueError("scale_max other than 448.0 (e4m3 max) is not supported")
    ops = ops_for(v)
    lib = _lib.load()
    shape, st = ops.shape(v), ops.strides(v)
    (B, H, S), s3 = _bhs(shape, st, tensor_layout)
    D = shape[3]
    code = ops.dtype_code(v)
    if code is None:
        raise ValueError("Input tensors must be in dtype of float16 or bfloat16")
    nbytes = lib.lbfa_v_fp8_bytes(B, H, S, D)
    buf = ops.empty((nbytes,), ops.uint8, v)
    v_scale = ops.empty((B, H, D), ops.float32, v)
    with ops.device_guard(v):
        _lib.check(lib.lbfa_quant_v_fp8(ops.ptr(v), code, ops.ptr(buf), ops.ptr(v_scale), B, H, S, D,
                                        _lib.strides3(s3), ops.stream(v)), lib)
    return Fp8V(buf, B, H, S, D), v_scale, None
