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
             seed=0, q_qmax=127, k_qmax=127, smooth_k=True):
    q, k, v = orc.make_inputs(B, H, S, D, seed=seed, layout=layout, dtype=dtype, Hkv=Hkv, Sk=Sk, k_bias=k_bias)
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
                  Sk=Sk or S, k_bias=k_bias, seed=seed, q_qmax=q_qmax, k_qmax=k_qmax, smooth_k=smooth_k)
    kmn = (km.float().numpy() if km is not None else np.zeros(0, np.float32))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), params=json.dumps(params), input_sha256=digest, km=kmn,
                        q_i8=q8.numpy(), k_i8=k8.numpy(), q_scale=q_scale.numpy(), k_scale=k_scale.numpy(),
                        o_bits=obits, lse2=lse2.numpy())
    print(f"{name}: o {tuple(o.shape)} |o|max {o.float().abs().max():.4f}")


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
]

if __name__ == "__main__":
    for c in CASES:
        run_case(**c)
