"""Host wrapper of the per-block quantiser for packed variable-length batches - same name, arguments and return
values as the reference's `src/triton/quant_per_block_varlen.py:75-142` (`per_block_int8`), launching the HIP
quantiser through the C ABI (`lbfa_quant_per_block_varlen`).

q [total_q, Hq, D], k [total_k, Hkv, D]; blocks restart at every sequence start; scales are
[sum_b ceil(len_b / BLK), H] with cu_seqlens_*_scale giving each sequence's first block (:92-106).
"""
from __future__ import annotations

from . import _lib
from ._tensor import ops_for


def cu_seqlens_scale(cu_seqlens, blk, ops):
    """`pad(cumsum((len + blk - 1) // blk), (1, 0))` (quant_per_block_varlen.py:92-100), on the device, int32."""
    lens = cu_seqlens[1:] - cu_seqlens[:-1]
    nblk = (lens + (blk - 1)) // blk
    return ops.cumsum0_pad(nblk)


def quantize(x, cu_seqlens, cu_scale, max_seqlen, *, sm_scale, qmax, blk, mean=None, total_blocks=None):
    """One launch (quant_per_block_varlen.py:107-123 / :125-141).  `mean`: optional [1, Hm, D] vector shared by all
    sequences (Hm divides H), subtracted in x's dtype before scaling.  Returns (codes int8 like x, scale
    fp32 [sum_blocks, H])."""
    ops = ops_for(x)
    lib = _lib.load()
    shape, st = ops.shape(x), ops.strides(x)
    total, H, D = shape
    if st[2] != 1:
        raise ValueError("Last dim of qkv must be contiguous.")
    code = ops.dtype_code(x)
    if code is None:
        raise ValueError("Input tensors must be in dtype of float16 or bfloat16")
    B = ops.shape(cu_seqlens)[0] - 1
    out = ops.empty(shape, ops.int8, x)
    ost = ops.strides(out)
    if total_blocks is None:
        total_blocks = int(cu_scale[-1])  # as the reference sizes the scale tensor (:101-106): one host read
    scale = ops.empty((total_blocks, H), ops.float32, x)
    mptr, mgroup = None, 1
    if mean is not None:
        ms = ops.shape(mean)
        hm = ms[-2]
        if hm <= 0 or H % hm != 0 or ms[-1] != D:
            raise ValueError(f"mean of shape {ms} does not match x of shape {shape}")
        mptr, mgroup = ops.ptr(mean), H // hm
    with ops.device_guard(x):
        _lib.check(lib.lbfa_quant_per_block_varlen(ops.ptr(x), code, mptr, mgroup, ops.ptr(out), ops.ptr(scale),
                                                   ops.ptr(cu_seqlens), ops.ptr(cu_scale), float(sm_scale), int(qmax),
                                                   int(blk), B, int(max_seqlen), H, D, _lib.strides2((st[1], st[0])),
                                                   _lib.strides2((ost[1], ost[0])), ops.stream(x)), lib)
    return out, scale


def per_block_int8(q, k, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, BLKQ=128, BLKK=64, sm_scale=None,
                   km=None):
    """Reference: src/triton/quant_per_block_varlen.py:75-142.  `km` (extension): the shared smoothing vector
    [1, Hkv, D]; the reference subtracts it in a separate pass before calling (src/core.py:452-454)."""
    ops = ops_for(q)
    head_dim = ops.shape(q)[-1]
    if sm_scale is None:
        sm_scale = head_dim ** -0.5  # :107-108
    cu_q, cu_k = ops.as_int32(cu_seqlens_q), ops.as_int32(cu_seqlens_k)
    cu_q_scale = cu_seqlens_scale(cu_q, BLKQ, ops)
    cu_k_scale = cu_seqlens_scale(cu_k, BLKK, ops)
    q_int8, q_scale = quantize(q, cu_q, cu_q_scale, max_seqlen_q, sm_scale=sm_scale * 1.44269504, qmax=127, blk=BLKQ)
    k_int8, k_scale = quantize(k, cu_k, cu_k_scale, max_seqlen_k, sm_scale=1.0, qmax=127, blk=BLKK, mean=km)
    return q_int8, q_scale, k_int8, k_scale, cu_q_scale, cu_k_scale
