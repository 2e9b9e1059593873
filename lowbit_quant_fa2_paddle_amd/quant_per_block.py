"""Host wrappers of the per-block quantiser - same names, arguments and return values as the
reference's `src/triton/quant_per_block.py` (`per_block_int8` :181-248, `per_block_int4_unpack`
:251-318, `per_block_q_int8_k_int4` :391-458), launching the HIP kernels through the C ABI.

Layouts are handled exactly as the reference does: by picking (batch, head, seq) strides
(:188-203) - no copies.  `k - km` is fused into the K launch instead of being a separate pass.
"""
from __future__ import annotations

from . import _lib
from ._tensor import ops_for


def _bhs(shape, strides, tensor_layout):
    """(B, H, S), (stride_b, stride_h, stride_s) for a [.., .., .., D] tensor (quant_per_block.py:188-203)."""
    if tensor_layout == "HND":
        return (shape[0], shape[1], shape[2]), (strides[0], strides[1], strides[2])
    if tensor_layout == "NHD":
        return (shape[0], shape[2], shape[1]), (strides[0], strides[2], strides[1])
    raise ValueError(f"Unknown tensor layout: {tensor_layout}")


def mean_seq(x, tensor_layout="HND"):
    """`x.mean(dim=seq_dim, keepdim=True)` (src/core.py:292-293) on the device: fp32 accumulate in a
    fixed order, one rounding to x's dtype.  Returns [B,H,1,D] (HND) or [B,1,H,D] (NHD)."""
    ops = ops_for(x)
    lib = _lib.load()
    shape, st = ops.shape(x), ops.strides(x)
    (B, H, S), s3 = _bhs(shape, st, tensor_layout)
    D = shape[3]
    code = ops.dtype_code(x)
    if code is None:
        raise ValueError("Input tensors must be in dtype of float16 or bfloat16")
    out = ops.empty((B, H, 1, D) if tensor_layout == "HND" else (B, 1, H, D), x.dtype, x)
    ws_bytes = lib.lbfa_mean_seq_workspace_bytes(B, H, S, D)
    ws = ops.empty((max(ws_bytes, 16) // 4,), ops.float32, x)
    with ops.device_guard(x):
        _lib.check(lib.lbfa_mean_seq(ops.ptr(x), code, ops.ptr(out), ops.ptr(ws), ws_bytes, B, H, S, D,
                                     _lib.strides3(s3), ops.stream(x)), lib)
    return out


def quantize(x, *, sm_scale, qmax, blk, tensor_layout="HND", mean=None, rowdot_vec=None):
    """One quantiser launch (quant_per_block.py:213-229 or :231-247).  `mean`: optional [B,Hm,1,D]/[B,1,Hm,D]
    tensor subtracted in x's dtype before scaling (Hm divides H).  `rowdot_vec`: optional vector of the same
    form; returns additionally fp32 [B,H,S] = dtype(x . vec) (the lse_correction of src/core.py:294-304).
    Returns (codes int8 like x, scale fp32 [B,H,ceil(S/blk)][, rowdot])."""
    ops = ops_for(x)
    lib = _lib.load()
    shape, st = ops.shape(x), ops.strides(x)
    (B, H, S), s3 = _bhs(shape, st, tensor_layout)
    D = shape[3]
    if st[3] != 1:
        raise ValueError("Last dim of qkv must be contiguous.")
    code = ops.dtype_code(x)
    if code is None:
        raise ValueError("Input tensors must be in dtype of float16 or bfloat16")
    out = ops.empty(shape, ops.int8, x)
    (_, _, _), o3 = _bhs(shape, ops.strides(out), tensor_layout)
    scale = ops.empty((B, H, (S + blk - 1) // blk), ops.float32, x)

    def _vec(v):
        if v is None:
            return None, 1
        vs = ops.shape(v)
        hm = vs[1] if tensor_layout == "HND" else vs[2]
        if hm <= 0 or H % hm != 0 or vs[3] != D:
            raise ValueError(f"mean/vector of shape {vs} does not match x of shape {shape}")
        return ops.ptr(v), H // hm

    mptr, mgroup = _vec(mean)
    vptr, vgroup = _vec(rowdot_vec)
    rowdot = ops.empty((B, H, S), ops.float32, x) if rowdot_vec is not None else None
    with ops.device_guard(x):
        _lib.check(lib.lbfa_quant_per_block(ops.ptr(x), code, mptr, mgroup, ops.ptr(out), ops.ptr(scale),
                                            float(sm_scale), int(qmax), int(blk), B, H, S, D,
                                            _lib.strides3(s3), _lib.strides3(o3), vptr, vgroup,
                                            ops.ptr(rowdot) if rowdot is not None else None, ops.stream(x)), lib)
    return (out, scale, rowdot) if rowdot_vec is not None else (out, scale)


def _per_block(q, k, km, BLKQ, BLKK, sm_scale, tensor_layout, q_qmax, k_qmax):
    head_dim = ops_for(q).shape(q)[3]
    if sm_scale is None:
        sm_scale = head_dim ** -0.5  # quant_per_block.py:210-211
    # Q: sm_scale * log2(e) folded into the codes' scale (:226); K: 1.0 (:244)
    q_int8, q_scale = quantize(q, sm_scale=sm_scale * 1.44269504, qmax=q_qmax, blk=BLKQ, tensor_layout=tensor_layout)
    k_int8, k_scale = quantize(k, sm_scale=1.0, qmax=k_qmax, blk=BLKK, tensor_layout=tensor_layout, mean=km)
    return q_int8, q_scale, k_int8, k_scale


def per_block_int8(q, k, km=None, BLKQ=128, BLKK=64, sm_scale=None, tensor_layout="HND"):
    """Reference: src/triton/quant_per_block.py:181-248."""
    return _per_block(q, k, km, BLKQ, BLKK, sm_scale, tensor_layout, 127, 127)


def per_block_int4_unpack(q, k, km=None, BLKQ=128, BLKK=64, sm_scale=None, tensor_layout="HND"):
    """Reference: src/triton/quant_per_block.py:251-318 - 4-bit range (+-7), one value per int8 byte."""
    return _per_block(q, k, km, BLKQ, BLKK, sm_scale, tensor_layout, 7, 7)


# The reference's packed `per_block_int4` (:321-388) writes half the rows and overlapping nibbles;
# the name resolves to the un-packed, correct definition.
per_block_int4 = per_block_int4_unpack


def per_block_q_int8_k_int4(q, k, km=None, BLKQ=128, BLKK=64, sm_scale=None, tensor_layout="HND"):
    """Intent of src/triton/quant_per_block.py:391-458: Q 8-bit (+-127), K 4-bit range (+-7), with the block
    sizes the attention kernel indexes scales by (the reference mis-sizes them)."""
    return _per_block(q, k, km, BLKQ, BLKK, sm_scale, tensor_layout, 127, 7)
