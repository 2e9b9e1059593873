"""Host wrapper of the fused attention kernel for packed variable-length batches - the `forward` of the
reference's `src/triton/attn_qk_int8_block_varlen.py:200-248` (non-causal) and
`src/triton/attn_qk_int8_per_block_causal_varlen.py:206-260` (causal), same arguments and return value.
"""
from __future__ import annotations

from . import _lib
from ._tensor import ops_for


def forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale,
            cu_seqlens_k_scale, output_dtype=None, is_causal=False, max_seqlen_k=None):
    """q [total_q, Hq, D], k [total_k, Hkv, D] int8 codes; v [total_k, Hkv, D] fp16 / bf16 (bf16 is converted to
    fp16 first, src/core.py:307-308); scales [sum_blocks, H] fp32 with their cu_seqlens_*_scale tables.
    `max_seqlen_k` (extension, default: total_k) only bounds the per-sequence address window check.
    Returns o [total_q, Hq, D] in `output_dtype` (default fp16)."""
    ops = ops_for(q)
    lib = _lib.load()
    qshape, kshape = ops.shape(q), ops.shape(k)
    _, Hq, D = qshape
    total_k, Hkv, _ = kshape
    B = ops.shape(cu_seqlens_q)[0] - 1
    if output_dtype is None:
        output_dtype = ops.float16
    o_code = _lib.LBFA_F16 if output_dtype == ops.float16 else (_lib.LBFA_BF16 if output_dtype == ops.bfloat16 else None)
    if o_code is None:
        raise ValueError("output_dtype must be float16 or bfloat16")
    v_code = ops.dtype_code(v)
    if v_code is None:
        raise ValueError("v must be float16 or bfloat16")
    o = ops.empty(qshape, output_dtype, q)
    hs = lambda t: _lib.strides2((ops.strides(t)[1], ops.strides(t)[0]))  # {head, token}
    if v_code == _lib.LBFA_BF16:  # the `v.to(float16)` of src/core.py:307-308, by the library's own kernel
        v16 = ops.empty(ops.shape(v), ops.float16, v)
        s3 = lambda t: _lib.strides3((0, ops.strides(t)[1], ops.strides(t)[0]))
        with ops.device_guard(q):
            _lib.check(lib.lbfa_cast_bf16_to_f16(ops.ptr(v), ops.ptr(v16), 1, Hkv, total_k, D, s3(v), s3(v16), ops.stream(q)), lib)
        v, v_code = v16, _lib.LBFA_F16
    with ops.device_guard(q):
        _lib.check(lib.lbfa_attn_fwd_varlen(ops.ptr(q), ops.ptr(k), ops.ptr(v), v_code, ops.ptr(o), o_code,
                                            ops.ptr(q_scale), ops.ptr(k_scale), ops.ptr(cu_seqlens_q), ops.ptr(cu_seqlens_k),
                                            ops.ptr(cu_seqlens_q_scale), ops.ptr(cu_seqlens_k_scale), B, Hq, Hkv,
                                            int(max_seqlen_q), int(max_seqlen_k if max_seqlen_k is not None else total_k), D,
                                            hs(q), hs(k), hs(v), hs(o), 1 if is_causal else 0, ops.stream(q)), lib)
    return o


def forward_causal(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale,
                   cu_seqlens_k_scale, output_dtype=None, max_seqlen_k=None):
    return forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale,
                   cu_seqlens_k_scale, output_dtype=output_dtype, is_causal=True, max_seqlen_k=max_seqlen_k)
