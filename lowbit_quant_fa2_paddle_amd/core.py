"""Public operator API - drop-in for the reference's `src/core.py` (names, arguments, defaults, error
behaviour), backed by hand-written gfx950 HIP kernels behind the C ABI of include/lowbit_fa.h.

Reference surface (src/__init__.py:1-17, aliases src/core.py:1099-1105):
    sageattn, sageattn_varlen, sageattn_qk_int8_pv_fp16_triton, sageattn_qk_int8_pv_fp16_cuda,
    sageattn_qk_int8_pv_fp8_cuda, sageattn_qk_int4_pv_fp16_triton, sageattn_multi_precision and the
    lowbit_fa_* aliases.
The `_triton` / `_cuda` suffixes name the reference's back ends; here every entry point runs the one
HIP back end (per-block scales, fp32 PV accumulation).  Back-end selectors (`quantization_backend`,
`qk_quant_gran`, `pv_accum_dtype`, `smooth_v`) are accepted and validated as the reference does
(src/core.py:320, :588-591, :721) and otherwise have no effect.
"""
from __future__ import annotations

import warnings
from typing import Any, Optional

from . import _lib
from . import quant_per_block as _qpb
from ._tensor import ops_for

_LOG2E = 1.44269504  # the literal of src/core.py:347


# ------------------------------------------------------------------------------------------------------
# shared prologue / epilogue (src/core.py:269-310, :343-350 - identical in all four reference operators)
# ------------------------------------------------------------------------------------------------------
def _check_inputs(q, k, v):
    ops = ops_for(q)
    dtype = q.dtype
    assert ops.dtype_code(q) is not None, "Input tensors must be in dtype of torch.float16 or torch.bfloat16"
    assert ops.same_device(q, k, v), "All tensors must be on the same device."
    assert q.dtype == k.dtype == v.dtype, "All tensors must have the same dtype."
    if not ops.is_gpu(q):
        raise RuntimeError("lowbit_fa operators run on an AMD GPU (gfx950) only; got a CPU tensor. "
                           "There is no CPU fallback.")
    return ops, dtype


def _pad_head_dim(ops, q, k, v):
    head_dim_og = ops.shape(q)[-1]
    if head_dim_og < 64:
        pad = 64 - head_dim_og
    elif 64 < head_dim_og < 128:
        pad = 128 - head_dim_og
    elif head_dim_og > 128:
        raise ValueError(f"Unsupported head_dim: {head_dim_og}")
    else:
        pad = 0
    # The one-call entry points take any head dim that is a multiple of 8 and treat channels up to 64 / 128 as the
    # zero padding of src/core.py:277-287 inside the kernels (never read, never written): no padded copies of q, k, v.
    # Other head dims (16-byte vector loads need rows of 8 elements) are padded here, as the reference does.
    if pad and head_dim_og % 8 == 0:
        pad = 0
    if pad:
        q, k, v = ops.pad_last(q, pad), ops.pad_last(k, pad), ops.pad_last(v, pad)
    assert ops.strides(q)[-1] == 1 and ops.strides(k)[-1] == 1 and ops.strides(v)[-1] == 1, \
        "Last dim of qkv must be contiguous."
    return q, k, v, head_dim_og


def _low_bit_attention(q, k, v, *, tensor_layout, is_causal, sm_scale, smooth_k, return_lse, q_qmax, k_qmax, pv):
    """The path of `sageattn_qk_int8_pv_fp16_triton` (src/core.py:269-352), parameterised on the code
    range of Q / K and the PV precision."""
    ops, dtype = _check_inputs(q, k, v)
    if tensor_layout not in ("HND", "NHD"):
        raise ValueError(f"Unknown tensor layout: {tensor_layout}")
    qshape, kshape = ops.shape(q), ops.shape(k)
    if 0 in qshape:  # nothing to compute: an empty output of the right shape (no launch)
        o = ops.empty(qshape, dtype, q)
        if return_lse:
            hdim = 1 if tensor_layout == "HND" else 2
            return o, ops.empty((qshape[0], qshape[hdim], qshape[3 - hdim]), ops.float32, q)
        return o
    if 0 in kshape:
        raise ValueError("k/v must hold at least one key (softmax over an empty set is undefined)")
    q, k, v, head_dim_og = _pad_head_dim(ops, q, k, v)
    if sm_scale is None:
        sm_scale = 1.0 / head_dim_og ** 0.5  # ORIGINAL head dim (:309-310)
    # mean_S(K), quantisation of Q and K (fp8: of V), attention and the LSE fix-up (:292-350) run inside ONE C-ABI
    # call on one caller-owned workspace: lbfa_forward == lbfa_mean_seq + 2 x lbfa_quant_per_block
    # (+ lbfa_quant_v_fp8) + lbfa_attn_fwd, which stay available (and tested) as separate entry points.
    lib = _lib.load()
    qshape, kshape = ops.shape(q), ops.shape(k)
    (B, Hq, Sq), q3 = _qpb._bhs(qshape, ops.strides(q), tensor_layout)
    (_, Hkv, Sk), k3 = _qpb._bhs(kshape, ops.strides(k), tensor_layout)
    (_, _, _), v3 = _qpb._bhs(ops.shape(v), ops.strides(v), tensor_layout)
    D = qshape[3]
    if is_causal and Sq != Sk:
        raise AssertionError("qo_len and kv_len must be equal for causal attention")  # causal forward :389
    o = ops.empty(qshape, dtype, q)
    (_, _, _), o3 = _qpb._bhs(qshape, ops.strides(o), tensor_layout)
    lse = ops.empty((B, Hq, Sq), ops.float32, q) if return_lse else None
    fp8 = 1 if pv == "fp8" else 0
    ws_bytes = lib.lbfa_forward_workspace_bytes_dt(B, Hq, Hkv, Sq, Sk, D, ops.dtype_code(q), fp8, 1 if smooth_k else 0, 1 if return_lse else 0)
    ws = ops.empty((ws_bytes,), ops.uint8, q)
    with ops.device_guard(q):
        _lib.check(lib.lbfa_forward(ops.ptr(q), ops.ptr(k), ops.ptr(v), ops.dtype_code(q), ops.ptr(o),
                                    ops.ptr(lse) if return_lse else None, ops.ptr(ws), ws_bytes,
                                    B, Hq, Hkv, Sq, Sk, D, _lib.strides3(q3), _lib.strides3(k3), _lib.strides3(v3),
                                    _lib.strides3(o3), float(sm_scale), int(q_qmax), int(k_qmax), fp8,
                                    1 if is_causal else 0, 1 if smooth_k else 0, ops.stream(q)), lib)
    o = o[..., :head_dim_og]
    return (o, lse) if return_lse else o


# ------------------------------------------------------------------------------------------------------
# operators
# ------------------------------------------------------------------------------------------------------
def sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout: str = "HND", quantization_backend: str = "triton",
                                    is_causal: bool = False, sm_scale: Optional[float] = None, smooth_k: bool = True,
                                    return_lse: bool = False, **kwargs: Any):
    """Per-block INT8 Q/K, FP16 PV (reference: src/core.py:194-352).

    q: [B, Hq, Sq, D] ("HND") or [B, Sq, Hq, D] ("NHD"); k, v likewise with Hkv | Hq; fp16 or bf16.
    Returns o (same shape/dtype as q) and, if `return_lse`, the natural-log LSE [B, Hq, Sq] fp32.
    """
    if quantization_backend not in ("triton", "cuda"):
        raise ValueError(f"Unsupported quantization backend: {quantization_backend}")  # :320
    return _low_bit_attention(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                              smooth_k=smooth_k, return_lse=return_lse, q_qmax=127, k_qmax=127, pv="fp16")


def sageattn_qk_int8_pv_fp16_cuda(q, k, v, tensor_layout: str = "HND", is_causal: bool = False,
                                  qk_quant_gran: str = "per_thread", sm_scale: Optional[float] = None,
                                  pv_accum_dtype: str = "fp32", smooth_k: bool = True, smooth_v: bool = False,
                                  return_lse: bool = False, **kwargs: Any):
    """INT8 Q/K, FP16 PV (reference: src/core.py:495-731).  `qk_quant_gran` and `pv_accum_dtype` pick among
    NVIDIA-fragment-specific variants in the reference; validated, then the per-block / fp32-accumulate
    HIP kernel runs."""
    assert qk_quant_gran in ["per_warp", "per_thread"], "qk_quant_gran must be either 'per_warp' or 'per_thread'."
    if pv_accum_dtype not in ("fp32", "fp16", "fp16+fp32"):
        raise ValueError(f"Unsupported pv_accum_dtype: {pv_accum_dtype}")  # :721
    if pv_accum_dtype in ["fp32", "fp16+fp32"] and smooth_v:
        warnings.warn(f"pv_accum_dtype is {pv_accum_dtype}, smooth_v will be ignored.")  # :642-644
    return _low_bit_attention(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                              smooth_k=smooth_k, return_lse=return_lse, q_qmax=127, k_qmax=127, pv="fp16")


def sageattn_qk_int8_pv_fp8_cuda(q, k, v, tensor_layout: str = "HND", is_causal: bool = False,
                                 qk_quant_gran: str = "per_thread", sm_scale: Optional[float] = None,
                                 pv_accum_dtype: str = "fp32+fp32", smooth_k: bool = True, smooth_v: bool = False,
                                 return_lse: bool = False, **kwargs: Any):
    """INT8 Q/K, FP8 (e4m3) P and V with per-channel V scales, fp32 accumulation
    (reference: src/core.py:735-941; kernel semantics csrc/qattn/qk_int_sv_f8_cuda.cu)."""
    assert qk_quant_gran in ["per_warp", "per_thread"], "qk_quant_gran must be either 'per_warp' or 'per_thread'."
    if pv_accum_dtype not in ("fp32", "fp32+fp32"):
        raise ValueError(f"Unsupported pv_accum_dtype: {pv_accum_dtype}")
    if pv_accum_dtype == "fp32+fp32" and smooth_v:
        warnings.warn("pv_accum_dtype is 'fp32+fp32', smooth_v will be ignored.")  # :879-881
    return _low_bit_attention(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                              smooth_k=smooth_k, return_lse=return_lse, q_qmax=127, k_qmax=127, pv="fp8")


def sageattn_qk_int4_pv_fp16_triton(q, k, v, tensor_layout: str = "HND", quantization_backend: str = "triton",
                                    is_causal: bool = False, sm_scale: Optional[float] = None, smooth_k: bool = True,
                                    return_lse: bool = False, q_bits: int = 4, **kwargs: Any):
    """4-bit-range Q/K codes, FP16 PV (reference API: src/core.py:945-1036).

    The reference body feeds a group-wise asymmetric packer's output to a kernel that expects per-block
    symmetric codes and cannot run; the arithmetic implemented is what its kernels define:
    `quant_per_block_int4_unpack_kernel` (+-7 codes, one per byte) + the int8 tile loop.
    `q_bits=4` (default): Q and K both +-7 ("qk_int4", what the function name and the int4 bench use);
    `q_bits=8`: Q +-127, K +-7 ("q_int8_k_int4", the bit widths the reference body passes, :999-1004).
    """
    if q_bits not in (4, 8):
        raise ValueError(f"q_bits must be 4 or 8, got {q_bits}")
    return _low_bit_attention(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                              smooth_k=smooth_k, return_lse=return_lse, q_qmax=7 if q_bits == 4 else 127, k_qmax=7,
                              pv="fp16")


def sageattn(q, k, v, tensor_layout: str = "HND", is_causal: bool = False, sm_scale: Optional[float] = None,
             return_lse: bool = False, **kwargs: Any):
    """Auto-dispatcher (reference: src/core.py:82-190 maps sm80/86/89/90 to a kernel family).  On gfx950 the
    headline INT8-QK / FP16-PV kernel is selected."""
    return sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal,
                                           sm_scale=sm_scale, return_lse=return_lse, **kwargs)


def sageattn_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q: int, max_seqlen_k: int,
                    is_causal: bool = False, sm_scale: Optional[float] = None, smooth_k: bool = True, **kwargs: Any):
    """Packed variable-length batch (reference: src/core.py:356-491): q [sum_q, Hq, D], k / v [sum_k, Hkv, D],
    cu_seqlens int32 / int64 [B+1] on the device.  As in the reference, smooth-K subtracts ONE mean taken over all
    packed tokens (`k.mean(dim=0)`, :452-454), quantisation blocks restart at every sequence, and each sequence
    attends only to itself (causal: len_q == len_k per sequence).  The whole batch runs in one `lbfa_forward_varlen`
    call - 4 launches whatever the number of sequences, no host read of cu_seqlens."""
    ops, dtype = _check_inputs(q, k, v)
    qshape = ops.shape(q)
    if len(qshape) != 3:
        raise ValueError("sageattn_varlen expects q of shape [total_q, num_qo_heads, head_dim]")
    if 0 in qshape:
        return ops.empty(qshape, dtype, q)
    if 0 in ops.shape(k):
        raise ValueError("k/v must hold at least one key (softmax over an empty set is undefined)")
    assert ops.shape(cu_seqlens_q) == ops.shape(cu_seqlens_k), "cu_seqlens_q and cu_seqlens_k must have the same length"
    q, k, v, head_dim_og = _pad_head_dim(ops, q, k, v)
    if sm_scale is None:
        sm_scale = 1.0 / head_dim_og ** 0.5  # :455-456
    cu_q, cu_k = ops.as_int32(cu_seqlens_q), ops.as_int32(cu_seqlens_k)
    lib = _lib.load()
    total_q, Hq, D = ops.shape(q)
    total_k, Hkv, _ = ops.shape(k)
    B = ops.shape(cu_q)[0] - 1
    o = ops.empty((total_q, Hq, D), dtype, q)
    hs = lambda t: _lib.strides2((ops.strides(t)[1], ops.strides(t)[0]))  # {head, token}
    ws_bytes = lib.lbfa_forward_varlen_workspace_bytes_dt(B, Hq, Hkv, total_q, total_k, int(max_seqlen_q), int(max_seqlen_k), D, ops.dtype_code(q))
    ws = ops.empty((max(ws_bytes, 16),), ops.uint8, q)
    with ops.device_guard(q):
        _lib.check(lib.lbfa_forward_varlen(ops.ptr(q), ops.ptr(k), ops.ptr(v), ops.dtype_code(q), ops.ptr(o),
                                           ops.ptr(cu_q), ops.ptr(cu_k), ops.ptr(ws), ws_bytes, B, Hq, Hkv, total_q, total_k,
                                           int(max_seqlen_q), int(max_seqlen_k), D, hs(q), hs(k), hs(v), hs(o),
                                           float(sm_scale), 127, 127, 1 if is_causal else 0, 1 if smooth_k else 0,
                                           ops.stream(q)), lib)
    return o[..., :head_dim_og]


def manual_scaled_dot_product_attention(q, k, v, is_causal=False):
    """The reference's FP16 fall-back (src/core.py:46-69) with the intended K^T (the `[0,2,3,1]` transpose at
    :55 is a defect).  Plain framework matmuls in the input dtype: this is the un-quantised
    branch of `sageattn_multi_precision`, not part of the low-bit hot path."""
    ops = ops_for(q)
    if ops.name != "torch":  # pragma: no cover
        import paddle
        scale = q.shape[-1] ** -0.5
        scores = paddle.matmul(q, k.transpose([0, 1, 3, 2])) * scale
        if is_causal:
            n = scores.shape[-1]
            scores = scores + (1 - paddle.tril(paddle.ones((n, n), dtype=scores.dtype))) * -1e9
        return paddle.matmul(paddle.nn.functional.softmax(scores, axis=-1), v)
    torch = ops.torch
    scale = q.shape[-1] ** -0.5
    scores = torch.matmul(q, k.transpose(-1, -2)) * scale
    if is_causal:
        n = scores.shape[-1]
        scores = scores + (1 - torch.tril(torch.ones((n, n), dtype=scores.dtype, device=scores.device))) * -1e9
    return torch.matmul(torch.softmax(scores, dim=-1), v)


def flash_attn_fp16(q, k, v, tensor_layout: str = "HND", is_causal: bool = False, sm_scale: Optional[float] = None,
                    return_lse: bool = False, **kwargs: Any):
    """Un-quantised FlashAttention-2 forward on the HIP back end (`lbfa_sdpa_fwd`): fp16 MFMAs for QK^T and PV, fp32
    softmax - the kernel behind the "FP16" branch of `sageattn_multi_precision` (the reference calls the framework's
    SDPA / `manual_scaled_dot_product_attention` there, src/core.py:46-69,1086-1087).  Same layouts, GQA, head dims
    and return convention as the low-bit operators; bf16 inputs are converted to fp16 inside the kernel."""
    ops, dtype = _check_inputs(q, k, v)
    if tensor_layout not in ("HND", "NHD"):
        raise ValueError(f"Unknown tensor layout: {tensor_layout}")
    qshape = ops.shape(q)
    if 0 in qshape:
        o = ops.empty(qshape, dtype, q)
        hdim = 1 if tensor_layout == "HND" else 2
        return (o, ops.empty((qshape[0], qshape[hdim], qshape[3 - hdim]), ops.float32, q)) if return_lse else o
    if 0 in ops.shape(k):
        raise ValueError("k/v must hold at least one key (softmax over an empty set is undefined)")
    q, k, v, head_dim_og = _pad_head_dim(ops, q, k, v)
    if sm_scale is None:
        sm_scale = 1.0 / head_dim_og ** 0.5
    lib = _lib.load()
    qshape = ops.shape(q)
    (B, Hq, Sq), q3 = _qpb._bhs(qshape, ops.strides(q), tensor_layout)
    (_, Hkv, Sk), k3 = _qpb._bhs(ops.shape(k), ops.strides(k), tensor_layout)
    (_, _, _), v3 = _qpb._bhs(ops.shape(v), ops.strides(v), tensor_layout)
    if is_causal and Sq != Sk:
        raise AssertionError("qo_len and kv_len must be equal for causal attention")
    o = ops.empty(qshape, dtype, q)
    (_, _, _), o3 = _qpb._bhs(qshape, ops.strides(o), tensor_layout)
    lse = ops.empty((B, Hq, Sq), ops.float32, q) if return_lse else None
    with ops.device_guard(q):
        _lib.check(lib.lbfa_sdpa_fwd(ops.ptr(q), ops.ptr(k), ops.ptr(v), ops.dtype_code(q), ops.ptr(o),
                                     ops.ptr(lse) if return_lse else None, B, Hq, Hkv, Sq, Sk, qshape[3],
                                     _lib.strides3(q3), _lib.strides3(k3), _lib.strides3(v3), _lib.strides3(o3),
                                     float(sm_scale), 1 if is_causal else 0, ops.stream(q)), lib)
    o = o[..., :head_dim_og]
    return (o, lse) if return_lse else o


default_attn = flash_attn_fp16


def _round_storage(x: float, code: int) -> float:
    """x rounded to fp16 / bf16 (round to nearest even), as an elementwise op of the framework on a tensor of that dtype rounds"""
    import struct
    if code == _lib.LBFA_F16:
        return struct.unpack("<e", struct.pack("<e", x))[0] if abs(x) < 65520.0 else (float("inf") if x > 0 else float("-inf"))
    bits = struct.unpack("<I", struct.pack("<f", x))[0]
    bits = (bits + 0x7FFF + ((bits >> 16) & 1)) & 0xFFFF0000
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def _absmax(tensors):
    """max |x| of each 4-D fp16 / bf16 tensor: one `lbfa_absmax` launch per tensor into one small device buffer, ONE copy to the
    host (the reference's `compute_scale` does a framework reduction per tensor and its comparisons synchronise, src/core.py:1039-1063)."""
    ops = ops_for(tensors[0])
    lib = _lib.load()
    out = ops.empty((len(tensors),), ops.float32, tensors[0])
    with ops.device_guard(tensors[0]):
        for i, t in enumerate(tensors):
            shp, st = ops.shape(t), ops.strides(t)
            if len(shp) != 4 or st[3] != 1 or ops.dtype_code(t) is None or not ops.is_gpu(t):
                raise ValueError("select_quantization: 4-D fp16 / bf16 device tensors with a contiguous last dim expected")
            _lib.check(lib.lbfa_absmax(ops.ptr(t), ops.dtype_code(t), ops.ptr(out) + 4 * i, shp[0], shp[1], shp[2], shp[3],
                                       _lib.strides3(st[:3]), ops.stream(t)), lib)
    return ops.tolist(out)


def compute_scale(tensor, bits=8, symmetric=True):
    """src/core.py:1039-1048: per-tensor scale max|x| / (2^(bits-1) - 1), in the tensor's dtype (the asymmetric form
    (max - min) / (2^bits - 1) is never used on the path: framework reductions, as the reference)."""
    if symmetric:
        code = ops_for(tensor).dtype_code(tensor)
        return _round_storage(_absmax([tensor])[0] / (2 ** (bits - 1) - 1), code)
    return float((tensor.max() - tensor.min()) / (2 ** bits - 1))


def select_quantization(q, k, v):
    """src/core.py:1051-1063: average per-tensor scale > 0.2 -> FP16, > 0.05 -> INT8, else INT4.  The three reductions run in
    the HIP library (`lbfa_absmax`); the scalar arithmetic behind them is done on the host, rounded to the tensors' dtype after
    every step as the reference's 0-d tensor arithmetic is."""
    code = ops_for(q).dtype_code(q)
    sq, sk, sv = (_round_storage(a / 127, code) for a in _absmax([q, k, v]))
    avg_scale = _round_storage(_round_storage(_round_storage(sq + sk, code) + sv, code) / 3.0, code)
    if avg_scale > 0.2:
        return "FP16"
    if avg_scale > 0.05:
        return "INT8"
    return "INT4"


def sageattn_multi_precision(q, k, v, tensor_layout: str = "HND", is_causal: bool = False,
                             sm_scale: Optional[float] = None, return_lse: bool = False, **kwargs: Any):
    """Importance-aware precision router (reference: src/core.py:1066-1096)."""
    kind = select_quantization(q, k, v)
    if kind == "FP16":
        return default_attn(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale, return_lse=return_lse)
    if kind == "INT8":
        return sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal,
                                               sm_scale=sm_scale, return_lse=return_lse)
    return sageattn_qk_int4_pv_fp16_triton(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal,
                                           sm_scale=sm_scale, return_lse=return_lse)


# Aliases with the preferred naming (src/core.py:1099-1105)
lowbit_fa_attn = sageattn
lowbit_fa_varlen = sageattn_varlen
lowbit_fa_multi_precision = sageattn_multi_precision
lowbit_fa_qk_int8_pv_fp16_triton = sageattn_qk_int8_pv_fp16_triton
lowbit_fa_qk_int8_pv_fp16_cuda = sageattn_qk_int8_pv_fp16_cuda
lowbit_fa_qk_int8_pv_fp8_cuda = sageattn_qk_int8_pv_fp8_cuda
lowbit_fa_qk_int4_pv_fp16_triton = sageattn_qk_int4_pv_fp16_triton
