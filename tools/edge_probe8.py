#!/usr/bin/env python3
"""Development probe: which strided operand breaks qk_int8_pv_fp16 at D = 128 (edge_probe7: fused HND views, every second token)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowbit_quant_fa2_paddle_amd as lb
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
def run(tag, tq, tk, tv, causal=False, smooth=True):
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, is_causal=causal, return_lse=True, smooth_k=smooth)
    o_ref, lse_ref = orc.lowbit_fa_forward(q, k, v, is_causal=causal, return_lse=True, amax_floor=1e-7, smooth_k=smooth, tail="neg_inf")
    on, ln = o.float().cpu().numpy(), lse.cpu().numpy()
    err = np.abs(on - o_ref) / (2e-3 + 2e-3 * np.abs(o_ref))
    bad = np.argwhere(err.max(-1) > 1)
    print(f"{tag}: worst err/tol {err.max():.2f} max|dLSE| {np.abs(ln - lse_ref).max():.2e} bad rows {len(bad)}/{err.shape[0]*err.shape[1]*err.shape[2]} first {bad[:4].tolist()} last {bad[-2:].tolist()}", flush=True)
g = torch.Generator(device=dev); g.manual_seed(7)
for D in (128, 64):
    B, H, S = 2, 4, 333
    c = lambda: torch.randn((B, H, S, D), generator=g, device=dev).half()
    def batch_strided(x):
        buf = torch.zeros((B, 3, H, S, D), device=dev, dtype=x.dtype); buf[:, 1] = x; return buf[:, 1]
    def token_strided(x):
        buf = torch.zeros((B, H, 2 * S, D), device=dev, dtype=x.dtype); buf[:, :, ::2] = x; return buf[:, :, ::2]
    def head_strided(x):
        buf = torch.zeros((B, 2 * H, S, D), device=dev, dtype=x.dtype); buf[:, ::2] = x; return buf[:, ::2]
    q, k, v = c(), c(), c()
    run(f"D{D} contiguous", q, k, v)
    for name, f in (("batch", batch_strided), ("token", token_strided), ("head", head_strided)):
        run(f"D{D} q {name}-strided", f(q), k, v)
        run(f"D{D} k {name}-strided", q, f(k), v)
        run(f"D{D} v {name}-strided", q, k, f(v))
    run(f"D{D} all token-strided, no smoothing", token_strided(q), token_strided(k), token_strided(v), smooth=False)
