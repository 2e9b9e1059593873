#!/usr/bin/env python3
"""Development probe: the un-quantised bf16 kernel on very large magnitudes - where do non-finite outputs appear?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lowbit_quant_fa2_paddle_amd import core
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
for mul in (1e5, 3e5, 1e6, 1e7):
    for D in (64, 128):
        for causal in (False, True):
            rng = np.random.default_rng(3)
            q, k, v = (rng.standard_normal((1, 2, 512, D)).astype(np.float32) for _ in range(3))
            q, k, v = orc.to_storage(q * mul, "bf16"), orc.to_storage(k * mul, "bf16"), orc.to_storage(v, "bf16")
            tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(dev) for x in (q, k, v))
            o, lse = core.flash_attn_fp16(tq, tk, tv, is_causal=causal, return_lse=True)
            on, ln = o.float().cpu().numpy(), lse.cpu().numpy()
            ref, rl = orc.sdpa_naive(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), is_causal=causal, return_lse=True)
            bo = ~np.isfinite(on); bl = ~np.isfinite(ln)
            msg = f"x{mul:g} D{D} causal={int(causal)}: O nonfinite rows {int(bo.any(-1).sum())}/1024 LSE nonfinite {int(bl.sum())}; |score| max ~{np.abs(rl).max():.2e} nat"
            if bo.any():
                rows = np.argwhere(bo.any(-1))[:6]
                msg += f"; first bad rows {rows.tolist()}; isnan {bool(np.isnan(on).any())} isinf {bool(np.isinf(on).any())}"
            print(msg, flush=True)
