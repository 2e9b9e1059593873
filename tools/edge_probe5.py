#!/usr/bin/env python3
"""Development probe: un-quantised kernels on large magnitudes - which sizes / dtypes / multipliers give non-finite outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lowbit_quant_fa2_paddle_amd import core
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
def run(dt, mul, S, D, causal, H=1):
    rng = np.random.default_rng(3)
    q, k, v = (rng.standard_normal((1, H, S, D)).astype(np.float32) for _ in range(3))
    q, k, v = orc.to_storage(q * mul, dt), orc.to_storage(k * mul, dt), orc.to_storage(v, dt)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to(tdt).to(dev) for x in (q, k, v))
    o, lse = core.flash_attn_fp16(tq, tk, tv, is_causal=causal, return_lse=True)
    on, ln = o.float().cpu().numpy(), lse.cpu().numpy()
    ref, rl = orc.sdpa_naive(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), is_causal=causal, return_lse=True)
    bo = ~np.isfinite(on); bl = ~np.isfinite(ln)
    print(f"{dt} x{mul:g} S{S} D{D} causal={int(causal)}: O nonfinite rows {int(bo.any(-1).sum())}/{H*S} (nan {int(np.isnan(on).any(-1).sum())}) LSE nan {int(np.isnan(ln).sum())} +inf {int((ln == np.inf).sum())} -inf {int((ln == -np.inf).sum())}; "
          f"max |lse| ref {np.abs(rl).max():.3g}; zero O rows {int((on == 0).all(-1).sum())}; first bad lse rows {np.argwhere(bl.reshape(-1))[:8, 0].tolist()}", flush=True)
for S in (64, 128, 256, 512):
    run("bf16", 1e5, S, 64, False)
for mul in (1e4, 2e4, 3e4, 4e4, 5e4, 7e4):
    run("bf16", mul, 512, 64, False)
for mul in (1e4, 3e4, 1e5):
    run("bf16", mul, 64, 64, False)
    run("bf16", mul, 64, 128, False)
