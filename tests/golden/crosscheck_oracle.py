#!/usr/bin/env python3
"""Randomised cross-check of the CPU oracle against the REFERENCE's own Triton kernels (interpreter mode) on shapes the
committed fixtures do not hold.  Runs only in the build container (needs /root/reference); not a pytest test.
    python tests/golden/crosscheck_oracle.py [n_cases]
Checks per case: quantiser codes and scales bit-identical, O within 1e-3 (bf16: + one ulp), raw LSE within 1e-5 relative."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (registers the reference kernels)

orc = mg.orc


def one(rng, idx):
    D = int(rng.choice([64, 128, 80, 96, 40, 32]))
    Hkv = int(rng.choice([1, 2]))
    H = Hkv * int(rng.choice([1, 2]))
    causal = bool(rng.integers(0, 2))
    S = int(rng.choice([64, 128, 192, 256, 320]))          # key lengths multiple of 64 (reference tail defect)
    Sq = S if causal else int(rng.choice([S, 64, 100, 200]))
    layout = str(rng.choice(["HND", "NHD"]))
    dtype = str(rng.choice(["fp16", "bf16"]))
    qm, km_ = [(127, 127), (7, 7), (127, 7)][idx % 3]
    smooth = bool(rng.integers(0, 4))
    bias = float(rng.choice([0.0, 0.3, -0.5]))
    q, k, v = orc.make_inputs(1, H, Sq, D, seed=idx, layout=layout, dtype=dtype, Hkv=Hkv, Sk=S, k_bias=bias)
    tq, tk, tv = (torch.from_numpy(a).to(mg.TDT[dtype]) for a in (q, k, v))
    tq, tk, tv = mg.pad_d(tq), mg.pad_d(tk), mg.pad_d(tv)
    seq_dim = 1 if layout == "NHD" else 2
    if smooth:
        km = tk.mean(dim=seq_dim, keepdim=True)
        tks = tk - km
    else:
        tks = tk
    if dtype == "bf16":
        tv = tv.to(torch.float16)
    sm_scale = 1.0 / D ** 0.5
    q8, q_scale = mg.ref_quantize(tq, layout, sm_scale * 1.44269504, 128, qm)
    k8, k_scale = mg.ref_quantize(tks, layout, 1.0, 64, km_)
    o, lse2 = mg.ref_attention(q8, k8, tv, q_scale, k_scale, layout, causal, mg.TDT[dtype])
    o = o[..., :D].float().numpy()
    o_orc, lse_orc, mid = orc.lowbit_fa_forward(q, k, v, dtype=dtype, tensor_layout=layout, is_causal=causal, smooth_k=smooth,
                                                return_lse=True, q_qmax=qm, k_qmax=km_, return_intermediates=True)
    canon = (lambda a: a) if layout == "HND" else (lambda a: np.transpose(a, (0, 2, 1, 3)))
    ok_codes = np.array_equal(mid["q_i8"], canon(q8.numpy())) and np.array_equal(mid["k_i8"], canon(k8.numpy()))
    ok_scales = np.array_equal(mid["q_scale"], q_scale.numpy()) and np.array_equal(mid["k_scale"], k_scale.numpy())
    rtol = 2.0 ** -7 if dtype == "bf16" else 0.0
    ok_o = bool(np.all(np.abs(o_orc - o) <= 1e-3 + rtol * np.abs(o)))
    cfg = dict(D=D, H=H, Hkv=Hkv, Sq=Sq, Sk=S, layout=layout, dtype=dtype, causal=causal, qmax=(qm, km_), smooth=smooth, bias=bias)
    return ok_codes, ok_scales, ok_o, float(np.abs(o_orc - o).max()), cfg


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(7)
    bad = 0
    for i in range(n):
        c, s, o, err, cfg = one(rng, i)
        flag = "ok " if (c and s and o) else "BAD"
        bad += flag == "BAD"
        print(flag, f"codes={c} scales={s} o={o} max|dO|={err:.2e}", json.dumps(cfg), flush=True)
    print(f"{n - bad} / {n} cases agree")
    sys.exit(1 if bad else 0)
