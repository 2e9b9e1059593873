#!/usr/bin/env python3
"""Development probe: fp8-PV and un-quantised operators on layout / dtype / GQA variants of the randint and huge-magnitude inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import core
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
TD = {"fp16": torch.float16, "bf16": torch.bfloat16}
for (H, Hkv, S, D, layout, dt, causal) in [(4, 2, 512, 128, "NHD", "fp16", True), (4, 2, 512, 64, "HND", "bf16", False), (2, 1, 333, 128, "HND", "bf16", True),
                                           (2, 2, 1000, 80, "NHD", "fp16", False), (6, 2, 256, 128, "HND", "fp16", False)]:
    q, k, v = orc.make_inputs(1, H, S, D, seed=7, layout=layout, dtype=dt, Hkv=Hkv, dist="randint")
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to(TD[dt]).to(dev) for x in (q, k, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp8_cuda(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True)
    o_ref, lse_ref = orc.lowbit_fa_forward(q, k, v, dtype=dt, tensor_layout=layout, is_causal=causal, return_lse=True, pv="fp8", amax_floor=1e-7)
    on = o.float().cpu().numpy()
    err = np.abs(on - o_ref)
    ulp = 2.0 ** -7 if dt == "bf16" else 0.0
    tol = 1e-2 + (2e-2 + ulp) * np.abs(o_ref)
    print(f"fp8 randint H{H}/{Hkv} S{S} D{D} {layout} {dt} causal={int(causal)}: finite {bool(np.isfinite(on).all())}, worst err/tol {np.max(err / tol):.2f}, max|dLSE| {np.abs(lse.cpu().numpy() - lse_ref).max():.2e} (|LSE| {np.abs(lse_ref).max():.1e})", flush=True)
# un-quantised kernels on huge bf16 magnitudes and randint
for (dt, mul, D, causal) in [("bf16", 1e4, 64, False), ("bf16", 1e6, 128, True), ("fp16", 60.0, 128, False), ("bf16", 1.0, 64, True)]:
    if mul == 1.0:
        q, k, v = orc.make_inputs(1, 2, 512, D, seed=9, dtype=dt, dist="randint")
    else:
        rng = np.random.default_rng(3)
        q, k, v = (rng.standard_normal((1, 2, 512, D)).astype(np.float32) for _ in range(3))
        q, k = orc.to_storage(q * mul, dt), orc.to_storage(k * mul, dt)
        v = orc.to_storage(v, dt)
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to(TD[dt]).to(dev) for x in (q, k, v))
    o, lse = core.flash_attn_fp16(tq, tk, tv, is_causal=causal, return_lse=True)
    ref, rl = orc.sdpa_naive(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), is_causal=causal, return_lse=True)
    on = o.float().cpu().numpy()
    tol = (2e-3 if dt == "fp16" else 4e-3) + (2e-3 + (2.0 ** -7 if dt == "bf16" else 0)) * np.abs(ref)
    fin = bool(np.isfinite(on).all() and np.isfinite(lse.cpu().numpy()).all())
    print(f"sdpa16 {dt} x{mul:g} D{D} causal={int(causal)}: finite {fin}, worst err/tol {np.max(np.abs(on - ref) / tol):.2f}, max rel dLSE {np.max(np.abs(lse.cpu().numpy() - rl) / (np.abs(rl) + 1)):.2e}", flush=True)
