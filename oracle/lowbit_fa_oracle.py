"""CPU oracle for the low-bit FlashAttention-2 forward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lowbit_quant_fa2_paddle_amd/`` may import this
module: it is the *checker* used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product path is the HIP library and fails loudly
when that library is missing.

What this file is: a numpy restatement of the reference's algorithm for the path
``lowbit_fa_qk_int{8,4}_pv_fp{16,8}_*`` (reference = Charles2530/lowbit_quant_fa2_paddle,
paths below are relative to that tree).  Every function cites the reference lines it
follows.  It is written for clarity and bit-level faithfulness, not speed; rows are
independent in the tile loop, so the loop over 64-key tiles is kept sequential (the online
softmax order matters) and everything else is vectorised over rows.

Parity pin: the int8 / fp16 functions are pinned against golden vectors produced by the
reference's own ``@triton.jit`` kernels executed under ``TRITON_INTERPRET=1`` (see
``tests/golden/make_golden.py`` and ``tests/test_oracle_golden.py``).  The fp8-PV functions
restate CUDA code that cannot be executed here (``csrc/qattn/qk_int_sv_f8_cuda.cu``) and
the reference holds no fixture for them: **parity unpinned** for fp8-PV, checked only
against fp32 SDPA within a stated tolerance.
"""
from __future__ import annotations

import math

import numpy as np

LOG2E = 1.44269504  # the literal the reference uses (src/triton/quant_per_block.py:226, src/core.py:347)
FP8_E4M3_MAX = 448.0
FP8_P_OFFSET = 8.807  # csrc/qattn/attn_utils.cuh:30  (log2(448) rounded) -> P_max = 448


# --------------------------------------------------------------------------------------
# dtype helpers
# --------------------------------------------------------------------------------------
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 -> bf16 (RNE) and return the value as fp32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    return np.where(np.isnan(x), x, out)


def to_storage(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round an fp32 array to the value set of `dtype` ('fp16'|'bf16'), returned as fp32."""
    if dtype == "fp16":
        return x.astype(np.float16).astype(np.float32)
    if dtype == "bf16":
        return bf16_round(x)
    raise ValueError(dtype)


def e4m3fn_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> OCP e4m3fn, round-to-nearest-even, saturate-to-finite (+-448); value as fp32.

    Follows the conversion the reference uses (`cvt.rn.satfinite.e4m3x2.f32`,
    csrc/numeric_conversion.cuh:39-54).  e4m3fn: 4 exponent bits (bias 7), 3 mantissa bits,
    min normal 2^-6, subnormal step 2^-9, max 448.
    """
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x).astype(np.float64)
    a = np.minimum(a, FP8_E4M3_MAX)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.maximum(e, -6.0)  # subnormals share the exponent of the smallest normal
    step = 2.0 ** (e - 3)
    q = np.rint(a / step) * step  # np.rint = round-half-even
    q = np.minimum(q, FP8_E4M3_MAX)
    return np.copysign(q, x).astype(np.float32)  # keeps the sign of a zero, as the hardware conversions do


def e4m3fn_encode(x: np.ndarray) -> np.ndarray:
    """Values already on the e4m3fn grid -> uint8 bit patterns."""
    x = np.asarray(x, dtype=np.float32)
    s = (np.signbit(x)).astype(np.uint8) << 7
    a = np.abs(x).astype(np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    nz = a > 0
    e = np.floor(np.log2(np.where(nz, a, 1.0)))
    e = np.maximum(e, -6.0)
    is_sub = a < 2.0 ** -6
    mant = np.where(is_sub, a / 2.0 ** -9, (a / 2.0 ** e - 1.0) * 8.0)
    ebits = np.where(is_sub, 0, e + 7).astype(np.int64)
    out = (ebits.astype(np.uint8) << 3) | np.rint(mant).astype(np.uint8)
    out = np.where(nz, out, 0).astype(np.uint8)
    return out | s


def e4m3fn_decode(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8)
    s = np.where(b & 0x80, -1.0, 1.0)
    e = ((b >> 3) & 0xF).astype(np.int64)
    m = (b & 0x7).astype(np.float64)
    v = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    return (s * v).astype(np.float32)


# --------------------------------------------------------------------------------------
# L2: per-block quantisation  (src/triton/quant_per_block.py)
# --------------------------------------------------------------------------------------
def quant_per_block(x: np.ndarray, sm_scale: float, qmax: float, blk: int, amax_floor: float = 0.0):
    """One tensor, canonical [B, H, S, C] fp32 values (already rounded to the storage dtype).

    Restates `quant_per_block_int8_kernel` (src/triton/quant_per_block.py:132-178, qmax=127)
    and `quant_per_block_int4_unpack_kernel` (:22-71, qmax=7):
        x = load(masked rows -> 0).to(fp32); x *= sm_scale; scale = max|x| / qmax;
        y = x / scale; y += 0.5*sign(y) (sign(0) := +1); int8(trunc(y)); one scale per block.
    `amax_floor`: the reference has no epsilon (:173) so an all-zero block yields scale=0 and
    NaN codes; the CUDA quantiser floors amax at 1e-7 (csrc/fused/fused.cu:147).  The HIP path
    adopts the floor; with the floor the result is bit-identical whenever amax >= 1e-7.
    Returns (codes int8 [B,H,S,C], scale fp32 [B,H,ceil(S/blk)]).
    """
    B, H, S, C = x.shape
    nblk = (S + blk - 1) // blk
    pad = nblk * blk - S
    xf = x.astype(np.float32)
    if pad:
        xf = np.concatenate([xf, np.zeros((B, H, pad, C), np.float32)], axis=2)
    xf = xf * np.float32(sm_scale)
    xb = xf.reshape(B, H, nblk, blk * C)
    amax = np.max(np.abs(xb), axis=-1)
    if amax_floor > 0.0:
        amax = np.maximum(amax, np.float32(amax_floor))
    scale = (amax / np.float32(qmax)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        y = xb / scale[..., None]
        y = y + np.float32(0.5) * np.where(y >= 0, np.float32(1), np.float32(-1))
        y = np.nan_to_num(y, nan=0.0)
    codes = np.trunc(y).astype(np.int32).astype(np.int8)
    codes = codes.reshape(B, H, nblk * blk, C)[:, :, :S]
    return np.ascontiguousarray(codes), scale


def mean_seq(k: np.ndarray, dtype: str) -> np.ndarray:
    """`km = k.mean(dim=seq, keepdim=True)` (src/core.py:292-293): fp32 accumulate, result
    rounded to the storage dtype.  k canonical [B,H,S,D] fp32-valued.  Returns [B,H,1,D]."""
    m = np.mean(k.astype(np.float64), axis=2, keepdims=True).astype(np.float32)
    return to_storage(m, dtype)


def per_block_int8(q, k, km, sm_scale, dtype: str, BLKQ=128, BLKK=64, q_qmax=127.0, k_qmax=127.0,
                   amax_floor: float = 0.0):
    """Host wrapper `per_block_int8` (src/triton/quant_per_block.py:181-248); with
    q_qmax=k_qmax=7 it is `per_block_int4_unpack` (:251-318); q_qmax=127,k_qmax=7 is the
    intent of `per_block_q_int8_k_int4` (:391-458, defective as written - SURVEY 2.4-5).

    `k - km` is an elementwise op in the storage dtype (:186-187) -> rounded to that dtype.
    Q gets sm_scale*1.44269504 folded in (:226), K uses 1.0 (:244).
    Canonical [B,H,S,D] arrays in / out.
    """
    if km is not None:
        k = to_storage(k.astype(np.float32) - km.astype(np.float32), dtype)
    # the product is formed in Python double and reaches the kernel as ONE fp32 argument (:226)
    q_i8, q_scale = quant_per_block(q, np.float32(float(sm_scale) * LOG2E), q_qmax, BLKQ, amax_floor)
    k_i8, k_scale = quant_per_block(k, 1.0, k_qmax, BLKK, amax_floor)
    return q_i8, q_scale, k_i8, k_scale


# --------------------------------------------------------------------------------------
# L1: fused attention tile loop  (src/triton/attn_qk_int8_per_block{,_causal}.py)
# --------------------------------------------------------------------------------------
def attn_fwd_int8_fp16(q_i8, k_i8, v16, q_scale, k_scale, *, causal: bool, out_dtype: str = "fp16",
                       return_lse: bool = False, BLOCK_M: int = 128, BLOCK_N: int = 64,
                       tail: str = "reference", pv_tile_fp16: bool = True):
    """`_attn_fwd` + `_attn_fwd_inner` (src/triton/attn_qk_int8_per_block.py:69-167 / :24-66) and the
    causal pair (src/triton/attn_qk_int8_per_block_causal.py:82-214 / :24-79).

    q_i8 [B,Hq,Sq,D] int8, k_i8 [B,Hkv,Sk,D] int8, v16 [B,Hkv,Sk,D] (fp16 values),
    q_scale [B,Hq,ceil(Sq/128)], k_scale [B,Hkv,ceil(Sk/64)] fp32.
    Per 64-key tile:  qk = int32(q.k) -> fp32 * q_scale * k_scale (:51);  m_ij = max(m_i, rowmax);
    p = exp2(qk - m_ij); l = l*alpha + sum(p); acc *= alpha (:52-58);  acc += fp16(dot(p.fp16, v))
    (:59-61, tile product rounded to fp16 when `pv_tile_fp16`, the reference's out_dtype=fp16);
    epilogue acc * (1/l) (:161-162), lse = log2(l) + m (:164-167, base-2 domain).
    Causal: tiles below the diagonal block unmasked (STAGE 1, causal :45-46), the 128x128 diagonal
    block gets `qk += where(row >= col, 0, -1e6)` (STAGE 2, :47-63); tiles above are never visited.
    `tail`: 'reference' = out-of-range K columns load as 0 so qk=0 takes part in the softmax
    (:48-49, the reference's ragged-tail defect); 'neg_inf' = masked with -inf (what
    the CUDA path does, csrc/qattn/attn_utils.cuh:327-353, and what the HIP kernel does).
    Rows are independent, so the loop runs over key tiles with all participating rows at once.
    Returns (o [B,Hq,Sq,D] fp32-valued rounded to out_dtype, lse [B,Hq,Sq] fp32 or None).
    """
    B, Hq, Sq, D = q_i8.shape
    _, Hkv, Sk, _ = k_i8.shape
    g = Hq // Hkv
    if causal:
        assert Sq == Sk, "qo_len and kv_len must be equal for causal attention"  # causal :389
    n_kt = (Sk + BLOCK_N - 1) // BLOCK_N
    o = np.zeros((B, Hq, Sq, D), np.float32)
    lse = np.zeros((B, Hq, Sq), np.float32) if return_lse else None
    rows = np.arange(Sq)
    row_blk = rows // BLOCK_M
    for b in range(B):
        for h in range(Hq):
            hk = h // g
            qf = q_i8[b, h].astype(np.float32)  # int8 products/sums < 2^24: exact in fp32
            kf = k_i8[b, hk].astype(np.float32)
            vf = v16[b, hk].astype(np.float16)
            qs = q_scale[b, h][row_blk].astype(np.float32)  # per-row copy of the per-block scale
            m_i = np.full(Sq, -np.inf, np.float32)
            l_i = np.ones(Sq, np.float32)  # :137  (erased by alpha=0 on the first tile)
            acc = np.zeros((Sq, D), np.float32)
            for j in range(n_kt):
                n0 = j * BLOCK_N
                n1 = min(n0 + BLOCK_N, Sk)
                if causal:
                    r0 = (n0 // BLOCK_M) * BLOCK_M  # q-tiles m with m*128 <= n0 take part
                    if r0 >= Sq:
                        break
                else:
                    r0 = 0
                kt = kf[n0:n1]
                if n1 - n0 < BLOCK_N:  # masked load -> zeros
                    kt = np.concatenate([kt, np.zeros((BLOCK_N - (n1 - n0), D), np.float32)], 0)
                qk = (qf[r0:] @ kt.T).astype(np.float32)
                qk = qk * qs[r0:, None] * k_scale[b, hk, j]
                if n1 - n0 < BLOCK_N and tail == "neg_inf":
                    qk[:, n1 - n0:] = -np.inf
                if causal:
                    # rows of the diagonal q-tile (first BLOCK_M participating rows) are masked
                    d1 = min(r0 + BLOCK_M, Sq)
                    rr = rows[r0:d1, None]
                    cc = (n0 + np.arange(BLOCK_N))[None, :]
                    qk[: d1 - r0] = qk[: d1 - r0] + np.where(rr >= cc, np.float32(0), np.float32(-1000000.0))
                m_ij = np.maximum(m_i[r0:], qk.max(axis=1))
                p = np.exp2(qk - m_ij[:, None]).astype(np.float32)
                l_ij = p.sum(axis=1, dtype=np.float32)
                alpha = np.exp2(m_i[r0:] - m_ij).astype(np.float32)
                l_i[r0:] = l_i[r0:] * alpha + l_ij
                acc[r0:] *= alpha[:, None]
                vt = vf[n0:n1]
                if n1 - n0 < BLOCK_N:
                    vt = np.concatenate([vt, np.zeros((BLOCK_N - (n1 - n0), D), np.float16)], 0)
                pv = p.astype(np.float16).astype(np.float32) @ vt.astype(np.float32)
                if pv_tile_fp16:
                    pv = pv.astype(np.float16).astype(np.float32)
                acc[r0:] += pv
                m_i[r0:] = m_ij
            o[b, h] = acc * (np.float32(1.0) / l_i)[:, None]
            if return_lse:
                lse[b, h] = np.log2(l_i) + m_i
    return to_storage(o, out_dtype), lse


# --------------------------------------------------------------------------------------
# fp8-PV variant (specification: csrc/qattn/qk_int_sv_f8_cuda.cu + csrc/fused/fused.cu) -- parity unpinned
# --------------------------------------------------------------------------------------
def per_channel_fp8(v: np.ndarray):
    """`per_channel_fp8` (src/quant.py:210-291) -> `MeanScaleKernel` (csrc/fused/fused.cu:317-428):
    per (b, h, d) channel: amax over tokens; scale = amax/448; v_fp8 = e4m3(v * 448/amax)
    (fused.cu:391-394,400,419-424).  v canonical [B,H,S,D] fp32-valued.
    Returns (v_fp8 values as fp32 [B,H,S,D], v_scale fp32 [B,H,D]).  The reference also
    transposes/pads/permutes V for its mma fragment (TransposePadPermuteKernel, fused.cu:263-314);
    that is a device layout choice, not arithmetic, and is not restated here.
    """
    amax = np.max(np.abs(v.astype(np.float32)), axis=2)  # [B,H,D]
    amax = np.maximum(amax, np.float32(1e-7))
    scale = (amax / np.float32(FP8_E4M3_MAX)).astype(np.float32)
    inv = (np.float32(FP8_E4M3_MAX) / amax).astype(np.float32)
    v8 = e4m3fn_round(v.astype(np.float32) * inv[:, :, None, :])
    return v8, scale


def attn_fwd_int8_fp8(q_i8, k_i8, v8, q_scale, k_scale, v_scale, *, causal: bool, out_dtype: str = "fp16",
                      return_lse: bool = False, BLOCK_M: int = 128, BLOCK_N: int = 64):
    """`qk_int_sv_f8_attn_kernel` semantics (csrc/qattn/qk_int_sv_f8_cuda.cu:46-692) on per-block
    scales: p = exp2(s - m + 8.807) so that p_max = 448 (csrc/qattn/attn_utils.cuh:30,424-445);
    l += sum(p) in fp32 before the fp8 rounding (:314-317); p -> e4m3 satfinite RN
    (attn_utils.cuh:470-485); PV e4m3 x e4m3 -> fp32 accumulate; O = (acc/l) * v_scale (:554-579);
    lse = log2(l) + m - 8.807 (:689).  Out-of-range keys are masked with -inf (attn_utils.cuh:327-353)
    and the causal mask is exact (-inf above the diagonal).
    """
    B, Hq, Sq, D = q_i8.shape
    _, Hkv, Sk, _ = k_i8.shape
    g = Hq // Hkv
    if causal:
        assert Sq == Sk
    n_kt = (Sk + BLOCK_N - 1) // BLOCK_N
    o = np.zeros((B, Hq, Sq, D), np.float32)
    lse = np.zeros((B, Hq, Sq), np.float32) if return_lse else None
    rows = np.arange(Sq)
    row_blk = rows // BLOCK_M
    off = np.float32(FP8_P_OFFSET)
    for b in range(B):
        for h in range(Hq):
            hk = h // g
            qf = q_i8[b, h].astype(np.float32)
            kf = k_i8[b, hk].astype(np.float32)
            vf = v8[b, hk].astype(np.float32)
            qs = q_scale[b, h][row_blk].astype(np.float32)
            m_i = np.full(Sq, -np.inf, np.float32)
            l_i = np.zeros(Sq, np.float32)
            acc = np.zeros((Sq, D), np.float32)
            for j in range(n_kt):
                n0 = j * BLOCK_N
                n1 = min(n0 + BLOCK_N, Sk)
                r0 = (n0 // BLOCK_M) * BLOCK_M if causal else 0
                if r0 >= Sq:
                    break
                qk = (qf[r0:] @ kf[n0:n1].T).astype(np.float32)
                qk = qk * qs[r0:, None] * k_scale[b, hk, j]
                if causal:
                    cc = (n0 + np.arange(n1 - n0))[None, :]
                    qk = np.where(rows[r0:, None] >= cc, qk, -np.inf)
                m_ij = np.maximum(m_i[r0:], qk.max(axis=1))
                p = np.exp2(qk - m_ij[:, None] + off).astype(np.float32)
                alpha = np.exp2(m_i[r0:] - m_ij).astype(np.float32)
                l_i[r0:] = l_i[r0:] * alpha + p.sum(axis=1, dtype=np.float32)
                acc[r0:] *= alpha[:, None]
                acc[r0:] += e4m3fn_round(p) @ vf[n0:n1]
                m_i[r0:] = m_ij
            o[b, h] = acc / l_i[:, None] * v_scale[b, hk][None, :]
            if return_lse:
                lse[b, h] = np.log2(l_i) + m_i - off
    return to_storage(o, out_dtype), lse


# --------------------------------------------------------------------------------------
# L3: the public operator  (src/core.py)
# --------------------------------------------------------------------------------------
def _canon(x: np.ndarray, tensor_layout: str) -> np.ndarray:
    """NHD [B,S,H,D] -> canonical [B,H,S,D].  The reference never copies: it swaps strides
    (src/triton/quant_per_block.py:188-203, src/triton/attn_qk_int8_per_block.py:183-196)."""
    if tensor_layout == "HND":
        return x
    if tensor_layout == "NHD":
        return np.transpose(x, (0, 2, 1, 3))
    raise ValueError(f"Unknown tensor layout: {tensor_layout}")


def _pad_head_dim(x: np.ndarray) -> np.ndarray:
    """src/core.py:277-287: D<64 -> 64, 64<D<128 -> 128 (zero pad), D>128 -> ValueError."""
    d = x.shape[-1]
    if d < 64:
        tgt = 64
    elif 64 < d < 128:
        tgt = 128
    elif d > 128:
        raise ValueError(f"Unsupported head_dim: {d}")
    else:
        return x
    pad = [(0, 0)] * (x.ndim - 1) + [(0, tgt - d)]
    return np.pad(x, pad)


def lowbit_fa_forward(q, k, v, *, dtype: str = "fp16", tensor_layout: str = "HND", is_causal: bool = False,
                      sm_scale=None, smooth_k: bool = True, return_lse: bool = False,
                      q_qmax: float = 127.0, k_qmax: float = 127.0, pv: str = "fp16",
                      tail: str = "reference", amax_floor: float = 0.0, pv_tile_fp16: bool = True,
                      return_intermediates: bool = False):
    """`sageattn_qk_int8_pv_fp16_triton` (src/core.py:194-352); with q_qmax/k_qmax = 7 the intended
    semantics of `sageattn_qk_int4_pv_fp16_triton` (:945-1036, body defective - SURVEY 2.4-2/3); with
    pv='fp8' `sageattn_qk_int8_pv_fp8_cuda` (:735-941) on per-block scales.

    q, k, v: fp32-valued numpy arrays already rounded to `dtype` ('fp16'|'bf16'), in `tensor_layout`.
    Steps: pad D (:277-287); km = mean_S(k) (:292-293); lse_correction = q.km (:294-304);
    bf16 -> v.half() (:307-308); sm_scale = D_og^-0.5 (:309-310); quant (:311-320); attention
    (:321-342); slice (:343); lse/1.44269504 + lse_correction*sm_scale (:344-350).
    """
    head_dim_og = q.shape[-1]
    q, k, v = _pad_head_dim(q), _pad_head_dim(k), _pad_head_dim(v)
    qc, kc, vc = _canon(q, tensor_layout), _canon(k, tensor_layout), _canon(v, tensor_layout)
    km = None
    lse_corr = None
    if smooth_k:
        km = mean_seq(kc, dtype)
        if return_lse:
            # matmul in the storage dtype, result rounded to it, then .to(float32) (:294-304)
            g = qc.shape[1] // kc.shape[1]
            kmq = np.repeat(km, g, axis=1) if g > 1 else km
            lse_corr = to_storage(np.einsum("bhsd,bhtd->bhs", qc.astype(np.float32), kmq.astype(np.float32)), dtype)
    v16 = vc.astype(np.float16).astype(np.float32)  # no-op for fp16 inputs (:307-308)
    if sm_scale is None:
        sm_scale = 1.0 / head_dim_og ** 0.5
    q_i8, q_scale, k_i8, k_scale = per_block_int8(qc, kc, km, sm_scale, dtype, q_qmax=q_qmax, k_qmax=k_qmax,
                                                  amax_floor=amax_floor)
    if pv == "fp16":
        o, lse = attn_fwd_int8_fp16(q_i8, k_i8, v16, q_scale, k_scale, causal=is_causal, out_dtype=dtype,
                                    return_lse=return_lse, tail=tail, pv_tile_fp16=pv_tile_fp16)
        v_scale = None
    elif pv == "fp8":
        v8, v_scale = per_channel_fp8(vc)
        o, lse = attn_fwd_int8_fp8(q_i8, k_i8, v8, q_scale, k_scale, v_scale, causal=is_causal, out_dtype=dtype,
                                   return_lse=return_lse)
    else:
        raise ValueError(pv)
    o = o[..., :head_dim_og]
    if tensor_layout == "NHD":
        o = np.ascontiguousarray(np.transpose(o, (0, 2, 1, 3)))
    if return_lse:
        lse = lse / np.float32(LOG2E)
        if smooth_k:
            lse = lse + lse_corr * np.float32(sm_scale)
    if return_intermediates:
        return o, lse, dict(q_i8=q_i8, q_scale=q_scale, k_i8=k_i8, k_scale=k_scale, km=km, v_scale=v_scale)
    return (o, lse) if return_lse else o


# --------------------------------------------------------------------------------------
# Packed variable-length batches  (src/core.py:356-491)
# --------------------------------------------------------------------------------------
def lowbit_fa_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, *, dtype: str = "fp16", is_causal: bool = False,
                     sm_scale=None, smooth_k: bool = True, q_qmax: float = 127.0, k_qmax: float = 127.0,
                     tail: str = "reference", amax_floor: float = 0.0, return_intermediates: bool = False):
    """`sageattn_varlen` (src/core.py:356-491).  q [total_q, Hq, D], k / v [total_k, Hkv, D] fp32-valued arrays
    already rounded to `dtype`; cu_seqlens_* integer sequences of B + 1 entries.

    Steps: pad D (:431-440); bf16 -> v.half() (:449-450); km = k.mean(dim=0) over ALL packed tokens, k - km in
    the storage dtype (:451-454); sm_scale = D_og^-0.5 (:455-456); per sequence, with blocks restarting at the
    sequence start (src/triton/quant_per_block_varlen.py:41-72): Q * sm_scale*1.44269504 -> int8 per 128 rows,
    K -> int8 per 64 rows; per sequence the same tile loop as the dense kernel
    (src/triton/attn_qk_int8_block_varlen.py:24-92, causal: attn_qk_int8_per_block_causal_varlen.py:24-84);
    slice (:490).  Returns o [total_q, Hq, D_og] (and, on request, the packed codes and the scales in the
    reference layout [sum_blocks, H])."""
    head_dim_og = q.shape[-1]
    q, k, v = _pad_head_dim(q), _pad_head_dim(k), _pad_head_dim(v)
    cq = [int(x) for x in cu_seqlens_q]
    ck = [int(x) for x in cu_seqlens_k]
    assert len(cq) == len(ck)
    v16 = v.astype(np.float16).astype(np.float32)
    km = None
    if smooth_k:
        km = to_storage(np.mean(k.astype(np.float64), axis=0, keepdims=True).astype(np.float32), dtype)  # [1,Hkv,D]
        k = to_storage(k.astype(np.float32) - km.astype(np.float32), dtype)
    if sm_scale is None:
        sm_scale = 1.0 / head_dim_og ** 0.5
    o = np.zeros(q.shape, np.float32)
    q_i8 = np.zeros(q.shape, np.int8)
    k_i8 = np.zeros(k.shape, np.int8)
    q_scales, k_scales = [], []
    for b in range(len(cq) - 1):
        qb = np.transpose(q[cq[b]:cq[b + 1]], (1, 0, 2))[None]   # [1,Hq,len,D]
        kb = np.transpose(k[ck[b]:ck[b + 1]], (1, 0, 2))[None]
        vb = np.transpose(v16[ck[b]:ck[b + 1]], (1, 0, 2))[None]
        if qb.shape[2] == 0:
            continue
        qi, qs = quant_per_block(qb, np.float32(float(sm_scale) * LOG2E), q_qmax, 128, amax_floor)
        q_i8[cq[b]:cq[b + 1]] = np.transpose(qi[0], (1, 0, 2))
        q_scales.append(np.transpose(qs[0], (1, 0)))             # [blocks, Hq]
        if kb.shape[2] == 0:
            continue  # l = 1, acc = 0 -> zeros (attn_qk_int8_block_varlen.py:171-173)
        ki, ks = quant_per_block(kb, 1.0, k_qmax, 64, amax_floor)
        k_i8[ck[b]:ck[b + 1]] = np.transpose(ki[0], (1, 0, 2))
        k_scales.append(np.transpose(ks[0], (1, 0)))
        ob, _ = attn_fwd_int8_fp16(qi, ki, vb, qs, ks, causal=is_causal, out_dtype=dtype, return_lse=False, tail=tail)
        o[cq[b]:cq[b + 1]] = np.transpose(ob[0], (1, 0, 2))
    o = o[..., :head_dim_og]
    if return_intermediates:
        return o, dict(q_i8=q_i8, k_i8=k_i8, km=km,
                       q_scale=np.concatenate(q_scales, axis=0) if q_scales else np.zeros((0, q.shape[1]), np.float32),
                       k_scale=np.concatenate(k_scales, axis=0) if k_scales else np.zeros((0, k.shape[1]), np.float32))
    return o


def make_varlen_inputs(lens_q, lens_k, Hq, Hkv, D, *, seed=0, dtype="fp16", k_bias=0.0):
    """Seeded packed inputs: q [sum(lens_q), Hq, D], k / v [sum(lens_k), Hkv, D] ~ N(0,1) rounded to `dtype`."""
    rng = np.random.default_rng(seed)
    tq, tk = int(sum(lens_q)), int(sum(lens_k))
    q = to_storage(rng.standard_normal((tq, Hq, D), dtype=np.float32), dtype)
    k = to_storage(rng.standard_normal((tk, Hkv, D), dtype=np.float32) + np.float32(k_bias), dtype)
    v = to_storage(rng.standard_normal((tk, Hkv, D), dtype=np.float32), dtype)
    cu_q = np.concatenate([[0], np.cumsum(lens_q)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(lens_k)]).astype(np.int32)
    return q, k, v, cu_q, cu_k


# --------------------------------------------------------------------------------------
# CPU baseline: the repo's naive SDPA  (src/core.py:46-69)
# --------------------------------------------------------------------------------------
def sdpa_naive(q, k, v, is_causal: bool = False, sm_scale=None, return_lse: bool = False):
    """`manual_scaled_dot_product_attention` (src/core.py:46-69) with the intended K^T
    (`transpose([0,1,3,2])`; the reference's `[0,2,3,1]` at :55 is a bug):
    scores = q.k^T * D^-0.5 (:55); causal: scores += (1 - tril) * -1e9 (:58-61); softmax (:64); @ v (:67).
    Canonical [B,H,S,D]; GQA handled by repeating kv heads.  fp32 math (float64 when inputs are)."""
    B, Hq, Sq, D = q.shape
    Hkv = k.shape[1]
    if Hkv != Hq:
        k = np.repeat(k, Hq // Hkv, axis=1)
        v = np.repeat(v, Hq // Hkv, axis=1)
    scale = D ** -0.5 if sm_scale is None else sm_scale
    scores = np.matmul(q, np.swapaxes(k, -1, -2)) * q.dtype.type(scale)
    if is_causal:
        Sk = scores.shape[-1]
        mask = np.tril(np.ones((Sq, Sk), dtype=scores.dtype))
        scores = scores + (1 - mask) * q.dtype.type(-1e9)
    mx = scores.max(axis=-1, keepdims=True)
    e = np.exp(scores - mx)
    s = e.sum(axis=-1, keepdims=True)
    out = np.matmul(e / s, v)
    if return_lse:
        return out, (np.log(s) + mx)[..., 0]
    return out


def sdpa_naive_torch(q, k, v, is_causal: bool = False, sm_scale=None):
    """The same `manual_scaled_dot_product_attention` (src/core.py:46-69) written with torch CPU tensor ops - the closest
    stand-in for the reference's Paddle-CPU path when Paddle is not installed (Paddle-CPU, else
    torch-CPU, else numpy).  q, k, v: torch CPU tensors [B,H,S,D].  Used for the CPU baseline timing only."""
    import torch
    D = q.shape[-1]
    scale = D ** -0.5 if sm_scale is None else sm_scale
    scores = torch.matmul(q, k.transpose(-1, -2)) * scale                      # :55 (intended K^T)
    if is_causal:
        Sq, Sk = scores.shape[-2], scores.shape[-1]
        mask = torch.tril(torch.ones((Sq, Sk), dtype=scores.dtype))            # :58-61
        scores = scores + (1 - mask) * -1e9
    return torch.matmul(torch.softmax(scores, dim=-1), v)                       # :64, :67


def attention_flops(B: int, H: int, Sq: int, Sk: int, D: int, causal: bool) -> float:
    """FLOP formula of the reference's bench harness: 4*B*H*D*S*S, halved for causal
    (utils/benchmark.py:212-214; example/test_sageattn_operator.py:96-98)."""
    f = 4.0 * B * H * D * Sq * Sk
    return f / 2 if causal else f


def make_inputs(B, H, S, D, *, seed=0, layout="HND", dtype="fp16", Hkv=None, Sk=None, k_bias=0.0, dist="normal"):
    """Seeded synthetic q,k,v: N(0,1) as example/test_sageattn_operator.py:43-52, or the bench
    distribution q,k = randint(-100,100), v ~ N(0,1) (utils/benchmark.py:215-230).  fp32-valued, rounded to dtype."""
    rng = np.random.default_rng(seed)
    Hkv = H if Hkv is None else Hkv
    Sk = S if Sk is None else Sk
    if dist == "normal":
        q = rng.standard_normal((B, H, S, D), dtype=np.float32)
        k = rng.standard_normal((B, Hkv, Sk, D), dtype=np.float32) + np.float32(k_bias)
    elif dist == "randint":
        q = rng.integers(-100, 100, (B, H, S, D)).astype(np.float32)
        k = rng.integers(-100, 100, (B, Hkv, Sk, D)).astype(np.float32) + np.float32(k_bias)
    else:
        raise ValueError(dist)
    v = rng.standard_normal((B, Hkv, Sk, D), dtype=np.float32)
    q, k, v = to_storage(q, dtype), to_storage(k, dtype), to_storage(v, dtype)
    if layout == "NHD":
        q, k, v = (np.ascontiguousarray(np.transpose(t, (0, 2, 1, 3))) for t in (q, k, v))
    return q, k, v
