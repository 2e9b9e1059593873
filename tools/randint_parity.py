#!/usr/bin/env python3
"""Measured distance of the HIP path from the reference on wide-score inputs (development / DESIGN record).

For every randint / peaky golden fixture (made by the reference's own Triton kernels, tests/golden/make_golden.py) and for larger
randint cases against the oracle: max |dO|, the number of elements and rows outside 2e-3 + 2e-3 |O|, max |dLSE| and its ratio
to |LSE|.  Run on the GPU box:  python tools/randint_parity.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import lowbit_quant_fa2_paddle_amd as lb
from conftest import golden_inputs, golden_names, load_golden
from lowbit_quant_fa2_paddle_amd import attn_qk_int8_per_block as attn
from oracle import lowbit_fa_oracle as orc

dev = torch.device("cuda:0")
TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


def stats(tag, o, ref, lse=None, lse_ref=None, dtype="fp16"):
    err = np.abs(o - ref)
    rtol = 2e-3 + (2.0 ** -7 if dtype == "bf16" else 0.0)
    bad = err > 2e-3 + rtol * np.abs(ref)
    msg = f"{tag:44s} max|dO| {err.max():.3e}  worst err/tol {np.max(err / (2e-3 + rtol * np.abs(ref))):.3f}  bad elems {int(bad.sum())}  bad rows {int(bad.any(axis=-1).sum())} of {bad[..., 0].size}"
    if lse is not None:
        le = np.abs(lse - lse_ref)
        msg += f"  max|dLSE| {le.max():.3e} (|LSE| max {np.abs(lse_ref).max():.3e}, ratio 2^{np.log2(max(le.max(), 1e-30) / np.abs(lse_ref).max()):.1f})"
    print(msg, flush=True)


for name in golden_names():
    p, g = load_golden(name)
    if p.get("dist", "normal") == "normal" and p.get("q_mul", 1.0) == 1.0:
        continue
    q, k, v = golden_inputs(orc, p)
    q8, k8 = torch.from_numpy(g["q_i8"]).to(dev), torch.from_numpy(g["k_i8"]).to(dev)
    qs, ks = torch.from_numpy(g["q_scale"]).to(dev), torch.from_numpy(g["k_scale"]).to(dev)
    tv = torch.from_numpy(np.ascontiguousarray(v)).to(TDT[p["dtype"]]).to(dev)
    o, lse = attn.forward(q8, k8, tv, qs, ks, tensor_layout=p["layout"], output_dtype=TDT[p["dtype"]], return_lse=True, is_causal=p["causal"])
    stats("kernel vs golden " + name, o.float().cpu().numpy()[..., :p["D"]], g["o"], lse.cpu().numpy(), g["lse2"], p["dtype"])

for (S, D, causal, seed) in [(1024, 64, False, 3), (1024, 64, True, 3), (1024, 128, False, 3), (1024, 128, True, 3),
                             (4096, 64, False, 5), (4096, 128, True, 6), (2048, 64, False, 7), (2048, 128, False, 8)]:
    q, k, v = orc.make_inputs(1, 2, S, D, seed=seed, dist="randint")
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).half().to(dev) for x in (q, k, v))
    for bits, fn, qm in ((8, lb.lowbit_fa_qk_int8_pv_fp16_triton, {}), (4, lb.lowbit_fa_qk_int4_pv_fp16_triton, dict(q_qmax=7, k_qmax=7))):
        o, lse = fn(tq, tk, tv, is_causal=causal, return_lse=True)
        o_ref, lse_ref = orc.lowbit_fa_forward(q, k, v, is_causal=causal, return_lse=True, amax_floor=1e-7, tail="neg_inf", **qm)
        stats(f"operator vs oracle randint int{bits} S{S} D{D}{' causal' if causal else ''}", o.float().cpu().numpy(), o_ref, lse.cpu().numpy(), lse_ref)
