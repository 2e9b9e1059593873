"""Host wrapper of the fused attention kernel - the `forward` of the reference's
`src/triton/attn_qk_int8_per_block.py:169-238` (non-causal) and
`src/triton/attn_qk_int8_per_block_causal.py:337-437` (causal), same arguments and returns.
"""
from __future__ import annotations

from . import _lib
from ._tensor import ops_for
from .quant import Fp8V
from .quant_per_block import _bhs


def forward(q, k, v, q_scale, k_scale, tensor_layout="HND", output_dtype=None, return_lse=False,
            is_causal=False, v_scale=None):
    """q, k: int8 codes; v: fp16 tensor in `tensor_layout` (a bf16 one is cast first, the `v.to(float16)` of
    src/core.py:307-308, by the library's own kernel), or an `Fp8V` (+ v_scale) for the fp8-PV kernel.  Returns (o, lse): o like q in `output_dtype` (default fp16), lse [B,Hq,Sq] fp32 in the
    kernel's base-2 domain, or an empty tensor when not requested (attn_qk_int8_per_block.py:201-204)."""
    ops = ops_for(q)
    lib = _lib.load()
    qshape = ops.shape(q)
    (B, Hq, Sq), q3 = _bhs(qshape, ops.strides(q), tensor_layout)
    (_, Hkv, Sk), k3 = _bhs(ops.shape(k), ops.strides(k), tensor_layout)
    D = qshape[3]
    if output_dtype is None:
        output_dtype = ops.float16
    o_code = _lib.LBFA_F16 if output_dtype == ops.float16 else (_lib.LBFA_BF16 if output_dtype == ops.bfloat16 else None)
    if o_code is None:
        raise ValueError("output_dtype must be float16 or bfloat16")
    if is_causal and Sq != Sk:
        raise AssertionError("qo_len and kv_len must be equal for causal attention")  # causal :389
    o = ops.empty(qshape, output_dtype, q)
    (_, _, _), o3 = _bhs(qshape, ops.strides(o), tensor_layout)
    if isinstance(v, Fp8V):
        if v_scale is None:
            raise ValueError("v_scale is required with fp8 V")
        if (v.B, v.H, v.S, v.D) != (B, Hkv, Sk, D):
            raise ValueError("fp8 V was quantised for a different shape")
        v_ptr, v_code, v3 = ops.ptr(v.buf), _lib.LBFA_E4M3, None
    else:
        v_code = ops.dtype_code(v)
        if v_code is None:
            raise ValueError("v must be float16 or bfloat16 (or the result of per_channel_fp8)")
        (_, _, _), v3s = _bhs(ops.shape(v), ops.strides(v), tensor_layout)
        if v_code == _lib.LBFA_BF16:
            v16 = ops.empty(ops.shape(v), ops.float16, v)
            (_, _, _), d3s = _bhs(ops.shape(v16), ops.strides(v16), tensor_layout)
            with ops.device_guard(q):
                _lib.check(lib.lbfa_cast_bf16_to_f16(ops.ptr(v), ops.ptr(v16), B, Hkv, Sk, D, _lib.strides3(v3s),
                                                     _lib.strides3(d3s), ops.stream(q)), lib)
            v, v3s, v_code = v16, d3s, _lib.LBFA_F16
        v_ptr, v3 = ops.ptr(v), _lib.strides3(v3s)
    lse = ops.empty((B, Hq, Sq), ops.float32, q) if return_lse else ops.empty((0,), ops.float32, q)
    with ops.device_guard(q):
        _lib.check(lib.lbfa_attn_fwd(ops.ptr(q), ops.ptr(k), v_ptr, v_code, ops.ptr(o), o_code,
                                     ops.ptr(lse) if return_lse else None, ops.ptr(q_scale), ops.ptr(k_scale),
                                     ops.ptr(v_scale) if v_scale is not None else None,
                                     B, Hq, Hkv, Sq, Sk, D, _lib.strides3(q3), _lib.strides3(k3), v3,
                                     _lib.strides3(o3), 1 if is_causal else 0, ops.stream(q)), lib)
    return o, lse


def forward_causal(q, k, v, q_scale, k_scale, tensor_layout="HND", output_dtype=None, return_lse=False, v_scale=None):
    return forward(q, k, v, q_scale, k_scale, tensor_layout=tensor_layout, output_dtype=output_dtype,
                   return_lse=return_lse, is_causal=True, v_scale=v_scale)
