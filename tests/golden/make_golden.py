#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own Triton kernels.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference's `@triton.jit` bodies are pure Triton, so they execute on CPU under
`TRITON_INTERPRET=1`.  Their files do `import paddle` at module level (paddle is not installed
here and stays absent): a module object carrying only the four dtype names the files evaluate at
import time (`paddle.int8/float16/bfloat16/float32` as default arguments) is registered so the
import statement succeeds.  No paddle functionality is emulated and none of the reference's
paddle host wrappers is called - their argument plumbing (stride selection, `k - km`, scale
shapes, launch grid) is restated below with torch-CPU tensors, citing the lines it follows, and
the reference kernels are launched directly:

    quant_per_block_int8_kernel          src/triton/quant_per_block.py:132-178
    quant_per_block_int4_unpack_kernel   src/triton/quant_per_block.py:22-71
    _attn_fwd (non-causal)               src/triton/attn_qk_int8_per_block.py:69-167
    _attn_fwd (causal)                   src/triton/attn_qk_int8_per_block_causal.py:82-214
    quant_per_block_int8_kernel (varlen) src/triton/quant_per_block_varlen.py:22-72
    _attn_fwd (varlen)                   src/triton/attn_qk_int8_block_varlen.py:94-197
    _attn_fwd (varlen, causal)           src/triton/attn_qk_int8_per_block_causal_varlen.py:83-203

Outputs: tests/golden/<case>.npz holding the case parameters, a checksum of the seeded inputs
(inputs are regenerated from the seed by oracle.make_inputs), km, int8 codes, scales, O (fp16/bf16
bit patterns) and the kernel's raw base-2 LSE.  Usage:  python tests/golden/make_golden.py
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import lowbit_fa_oracle as orc  # noqa: E402  (input generator + bf16 helpers only)

_stub = types.ModuleType("paddle")
_stub.int8, _stub.float16, _stub.bfloat16, _stub.float32 = torch.int8, torch.float16, torch.bfloat16, torch.float32
sys.modules.setdefault("paddle", _stub)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_quant = _load("ref_quant_per_block", "src/triton/quant_per_block.py")
ref_attn = _load("ref_attn_noncausal", "src/triton/attn_qk_int8_per_block.py")
ref_attn_c = _load("ref_attn_causal", "src/triton/attn_qk_int8_per_block_causal.py")

ref_quant_vl = _load("ref_quant_per_block_varlen", "src/triton/quant_per_block_varlen.py")
ref_attn_vl = _load("ref_attn_varlen", "src/triton/attn_qk_int8_block_varlen.py")
ref_attn_vl_c = _load("ref_attn_varlen_causal", "src/triton/attn_qk_int8_per_block_causal_varlen.py")

TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


def _strides(t, layout):
    # src/triton/quant_per_block.py:188-203 / src/triton/attn_qk_int8_per_block.py:183-196
    s = t.stride()
    return (s[0], s[1], s[2]) if layout == "HND" else (s[0], s[2], s[1])


def _dims(t, layout):
    return (t.shape[0], t.shape[1], t.shape[2]) if layout == "HND" else (t.shape[0], t.shape[2], t.shape[1])


def ref_quantize(x, layout, sm_scale, blk, qmax):
    """One launch as src/triton/quant_per_block.py:212-229 (int8) / :282-299 (int4_unpack)."""
    b, h, L = _dims(x, layout)
    C = x.shape[-1]
    out = torch.empty(x.shape, dtype=torch.int8)
    scale = torch.empty((b, h, (L + blk - 1) // blk), dtype=torch.float32)
    kern = ref_quant.quant_per_block_int8_kernel if qmax == 127 else ref_quant.quant_per_block_int4_unpack_kernel
    grid = ((L + blk - 1) // blk, h, b)
    kern[grid](x, out, scale, L, *_strides(x, layout), *_strides(out, layout), scale.stride(0), scale.stride(1),
               sm_scale=sm_scale, C=C, BLK=blk)
    return out, scale


def ref_attention(q8, k8, v, q_scale, k_scale, layout, causal, out_dtype, return_lse=True):
    """Launch as src/triton/attn_qk_int8_per_block.py:205-237 / ..._causal.py:396-436."""
    b, hq, Sq = _dims(q8, layout)
    _, hkv, Sk = _dims(k8, layout)
    D = q8.shape[-1]
    o = torch.empty(q8.shape, dtype=out_dtype)
    lse = torch.empty((b, hq, Sq), dtype=torch.float32)
    grid = ((Sq + 127) // 128, hq, b)
    common = (*_strides(q8, layout), *_strides(k8, layout), *_strides(v, layout), *_strides(o, layout))
    if not causal:
        ref_attn._attn_fwd[grid](q8, k8, v, q_scale, k_scale, o, lse, *common, Sq, Sk,
                                 H=hq, num_kv_groups=hq // hkv, HEAD_DIM=D, BLOCK_M=128, BLOCK_N=64,
                                 STAGE=1, RETURN_LSE=return_lse)
    else:
        # o_scale / o_mn are written per ROW by the kernel (causal :209-214); the reference host allocates
        # them per block (:351-352, out of bounds).  Allocate per row so the stores land in bounds.
        o_scale = torch.empty((b, hq, Sq), dtype=torch.float32)
        o_mn = torch.empty((b, hq, Sq), dtype=torch.float32)
        ref_attn_c._attn_fwd[grid](q8, k8, v, q_scale, k_scale, o, lse, o_scale, o_mn, *common,
                                   *o_scale.stride(), *o_mn.stride(), Sq, Sk, hq, hq // hkv,
                                   BLOCK_M=128, BLOCK_N=64, HEAD_DIM=D, STAGE=3, RETURN_LSE=return_lse)
    return o, lse


def pad_d(t):
    d = t.shape[-1]
    tgt = 64 if d < 64 else (128 if 64 < d < 128 else d)
    return torch.nn.functional.pad(t, (0, tgt - d)) if tgt != d else t


def run_case(name, B, H, S, D, layout="HND", causal=False, dtype="fp16", Hkv=None, Sk=None, k_bias=0.0,
             seed=0, q_qmax=127, k_qmax=127, smooth_k=True, dist="normal", q_mul=1.0):
    q, k, v = orc.make_inputs(B, H, S, D, seed=seed, layout=layout, dtype=dtype, Hkv=Hkv, Sk=Sk, k_bias=k_bias, dist=dist)
    if q_mul != 1.0:
        q = orc.to_storage(q * np.float32(q_mul), dtype)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (q, k, v))).hexdigest()
    tq, tk, tv = (torch.from_numpy(a).to(TDT[dtype]) for a in (q, k, v))
    head_dim_og = D
    tq, tk, tv = pad_d(tq), pad_d(tk), pad_d(tv)  # src/core.py:277-287
    seq_dim = 1 if layout == "NHD" else 2
    if smooth_k:
        km = tk.mean(dim=seq_dim, keepdim=True)  # src/core.py:292-293
        tks = tk - km  # src/triton/quant_per_block.py:186-187 (storage-dtype elementwise)
    else:
        km, tks = None, tk
    if dtype == "bf16":
        tv = tv.to(torch.float16)  # src/core.py:307-308
    sm_scale = 1.0 / head_dim_og ** 0.5  # src/core.py:309-310
    q8, q_scale = ref_quantize(tq, layout, sm_scale * 1.44269504, 128, q_qmax)
    k8, k_scale = ref_quantize(tks, layout, 1.0, 64, k_qmax)
    o, lse2 = ref_attention(q8, k8, tv, q_scale, k_scale, layout, causal, TDT[dtype])
    o = o[..., :head_dim_og]
    obits = o.contiguous().view(torch.int16).numpy().view(np.uint16)
    params = dict(name=name, B=B, H=H, S=S, D=D, layout=layout, causal=causal, dtype=dtype, Hkv=Hkv or H,
                  Sk=Sk or S, k_bias=k_bias, seed=seed, q_qmax=q_qmax, k_qmax=k_qmax, smooth_k=smooth_k, dist=dist, q_mul=q_mul)
    kmn = (km.float().numpy() if km is not None else np.zeros(0, np.float32))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), params=json.dumps(params), input_sha256=digest, km=kmn,
                        q_i8=q8.numpy(), k_i8=k8.numpy(), q_scale=q_scale.numpy(), k_scale=k_scale.numpy(),
                        o_bits=obits, lse2=lse2.numpy())
    print(f"{name}: o {tuple(o.shape)} |o|max {o.float().abs().max():.4f}")


def ref_quantize_varlen(x, cu, max_len, sm_scale, blk):
    """One launch as src/triton/quant_per_block_varlen.py:86-124 (scale layout [sum_blocks, H], :92-106)."""
    h, C = x.shape[1], x.shape[2]
    b = cu.shape[0] - 1
    lens = cu[1:] - cu[:-1]
    cu_scale = torch.nn.functional.pad(torch.cumsum((lens + blk - 1) // blk, dim=0), (1, 0), value=0).to(torch.int32)
    out = torch.empty(x.shape, dtype=torch.int8)
    scale = torch.empty((int(cu_scale[-1]), h), dtype=torch.float32)
    grid = ((max_len + blk - 1) // blk, h, b)
    ref_quant_vl.quant_per_block_int8_kernel[grid](x, out, scale, cu, cu_scale, x.stride(1), x.stride(0), out.stride(1),
                                                   out.stride(0), sm_scale=sm_scale, H=h, C=C, BLK=blk)
    return out, scale, cu_scale


def run_varlen_case(name, lens_q, lens_k, Hq, Hkv, D, causal=False, dtype="fp16", k_bias=0.0, seed=0, smooth_k=True):
    q, k, v, cu_q, cu_k = orc.make_varlen_inputs(lens_q, lens_k, Hq, Hkv, D, seed=seed, dtype=dtype, k_bias=k_bias)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (q, k, v))).hexdigest()
    tq, tk, tv = (torch.from_numpy(a).to(TDT[dtype]) for a in (q, k, v))
    tcq, tck = torch.from_numpy(cu_q), torch.from_numpy(cu_k)
    head_dim_og = D
    tq, tk, tv = pad_d(tq), pad_d(tk), pad_d(tv)  # src/core.py:431-440
    if dtype == "bf16":
        tv = tv.to(torch.float16)  # :449-450
    if smooth_k:
        km = tk.mean(dim=0, keepdim=True)  # :452-454
        tk = tk - km
    else:
        km = None
    sm_scale = 1.0 / head_dim_og ** 0.5
    max_q, max_k = int(max(lens_q)), int(max(lens_k))
    q8, q_scale, cu_qs = ref_quantize_varlen(tq, tcq, max_q, sm_scale * 1.44269504, 128)
    k8, k_scale, cu_ks = ref_quantize_varlen(tk, tck, max_k, 1.0, 64)
    o = torch.zeros(tq.shape, dtype=TDT[dtype])
    b = len(lens_q)
    grid = ((max_q + 127) // 128, Hq, b)
    mod = ref_attn_vl_c if causal else ref_attn_vl
    # launch as attn_qk_int8_block_varlen.py:216-247 / attn_qk_int8_per_block_causal_varlen.py:228-259
    mod._attn_fwd[grid](q8, k8, tv, tcq, tck, q_scale, k_scale, cu_qs, cu_ks, o, tq.stride(1), tq.stride(0), k8.stride(1),
                        k8.stride(0), tv.stride(1), tv.stride(0), o.stride(1), o.stride(0), Hq, Hq // Hkv,
                        BLOCK_M=128, BLOCK_N=64, HEAD_DIM=tq.shape[-1], STAGE=3 if causal else 1)
    o = o[..., :head_dim_og]
    obits = o.contiguous().view(torch.int16).numpy().view(np.uint16)
    params = dict(name=name, lens_q=list(map(int, lens_q)), lens_k=list(map(int, lens_k)), Hq=Hq, Hkv=Hkv, D=D, causal=causal,
                  dtype=dtype, k_bias=k_bias, seed=seed, smooth_k=smooth_k)
    kmn = (km.float().numpy() if km is not None else np.zeros(0, np.float32))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), params=json.dumps(params), input_sha256=digest, km=kmn,
                        q_i8=q8.numpy(), k_i8=k8.numpy(), q_scale=q_scale.numpy(), k_scale=k_scale.numpy(),
                        cu_q_scale=cu_qs.numpy(), cu_k_scale=cu_ks.numpy(), o_bits=obits)
    print(f"{name}: o {tuple(o.shape)} |o|max {o.float().abs().max():.4f}")


# K lengths are multiples of 64: for ragged key tails the reference kernel lets the padded keys into the softmax
# as zeros, which the HIP path deliberately does not reproduce.  Query lengths are ragged.
VARLEN_CASES = [
    dict(name="varlen_h4_kv2_d64", lens_q=[128, 200, 64], lens_k=[192, 256, 64], Hq=4, Hkv=2, D=64, k_bias=0.3, seed=11),
    dict(name="varlen_h4_kv2_d64_causal", lens_q=[192, 256, 64], lens_k=[192, 256, 64], Hq=4, Hkv=2, D=64, k_bias=0.3,
         seed=11, causal=True),
    dict(name="varlen_bf16_d128", lens_q=[70, 130], lens_k=[128, 64], Hq=2, Hkv=2, D=128, dtype="bf16", seed=12),
]

CASES = [
    dict(name="c1_hnd_s256_d64", B=1, H=2, S=256, D=64),  # BASELINE config 1
    dict(name="c1_hnd_s256_d64_causal", B=1, H=2, S=256, D=64, causal=True),
    dict(name="nhd_s512_d128_kbias", B=1, H=2, S=512, D=128, layout="NHD", k_bias=0.5, seed=1),
    dict(name="nhd_s512_d128_kbias_causal", B=1, H=2, S=512, D=128, layout="NHD", k_bias=0.5, seed=1, causal=True),
    dict(name="gqa_h4_kv2_s256_d64", B=2, H=4, Hkv=2, S=256, D=64, seed=2),
    dict(name="gqa_h4_kv2_s256_d64_causal", B=2, H=4, Hkv=2, S=256, D=64, seed=2, causal=True),
    dict(name="bf16_s256_d64", B=1, H=2, S=256, D=64, dtype="bf16", seed=3),
    dict(name="pad_d80_s256", B=1, H=2, S=256, D=80, seed=4),
    dict(name="cross_sq128_sk320_d64", B=1, H=2, S=128, Sk=320, D=64, seed=5),
    dict(name="int4_s256_d64", B=1, H=2, S=256, D=64, seed=6, q_qmax=7, k_qmax=7),
    dict(name="int4_s256_d64_causal", B=1, H=2, S=256, D=64, seed=6, q_qmax=7, k_qmax=7, causal=True),
    dict(name="q8k4_s256_d128", B=1, H=2, S=256, D=128, seed=7, q_qmax=127, k_qmax=7),
    dict(name="nosmooth_s256_d64", B=1, H=2, S=256, D=64, seed=8, smooth_k=False),
    # 4-bit-range codes with a padded head dim: pins fp32(sm_scale * 1.44269504) formed in double (a float-by-float
    # product is one ulp off for D = 80 / 96 / 40 and flips codes)
    dict(name="int4_pad_d80_bf16_s256", B=1, H=2, S=256, D=80, dtype="bf16", seed=9, q_qmax=7, k_qmax=7),
    dict(name="int4_pad_d96_s256_causal", B=1, H=2, S=256, D=96, seed=10, q_qmax=7, k_qmax=7, causal=True),
    # the reference's own benchmark distribution q, k = randint(-100, 100), v ~ N(0,1) (utils/benchmark.py:215-230): scores
    # thousands of binades apart, one-hot rows - the inputs on which a lazy softmax reference has to fall back (round 3)
    dict(name="randint_s512_d64", B=1, H=2, S=512, D=64, seed=14, dist="randint"),
    dict(name="randint_s512_d64_causal", B=1, H=2, S=512, D=64, seed=14, dist="randint", causal=True),
    dict(name="randint_int4_s384_d128", B=1, H=2, S=384, D=128, seed=15, dist="randint", q_qmax=7, k_qmax=7),
    # peaky N(0,1): queries x 6 (scores ~8 binades wide) - some rows outgrow the first key tile's maximum by more than 2^16
    # in a later tile, others never do: the mixed case of the lazy / exact machinery, 8..12 key tiles
    dict(name="peaky_q6_s768_d64", B=1, H=2, S=768, D=64, seed=16, q_mul=6.0, k_bias=0.3),
    dict(name="peaky_q6_s512_d128_nhd_causal", B=1, H=2, S=512, D=128, seed=17, q_mul=6.0, layout="NHD", causal=True),
    # round 4: the bench distribution on 8-bit codes at D = 128 with GQA, NHD and the causal mask (C3's shape family), and with
    # mixed code ranges (q int8, k 4-bit range)
    dict(name="randint_s384_d128_nhd_gqa_causal", B=1, H=4, Hkv=2, S=384, D=128, seed=18, dist="randint", layout="NHD", causal=True),
    dict(name="randint_q8k4_s320_d64", B=1, H=2, S=320, D=64, seed=19, dist="randint", q_qmax=127, k_qmax=7),
]

if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for c in CASES:
        if only in c["name"]:
            run_case(**c)
    for c in VARLEN_CASES:
        if only in c["name"]:
            run_varlen_case(**c)
