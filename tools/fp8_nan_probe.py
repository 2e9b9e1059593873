#!/usr/bin/env python3
"""Development probe: fp8-PV operator on the reference's randint distribution - non-finite outputs and distance from the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowbit_quant_fa2_paddle_amd as lb
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
# the soak's failing configuration
g = torch.Generator(device=dev); g.manual_seed(1234)
B, H, S, D = 1, 8, 2048, 128
q = torch.randint(-100, 100, (B, H, S, D), generator=g, device=dev).half()
k = torch.randint(-100, 100, (B, H, S, D), generator=g, device=dev).half()
v = torch.randn((B, H, S, D), generator=g, device=dev).half()
for lse_flag in (True, False):
    r = lb.lowbit_fa_qk_int8_pv_fp8_cuda(q, k, v, return_lse=lse_flag)
    o = r[0] if lse_flag else r
    print("soak config return_lse", lse_flag, "O nonfinite", int((~torch.isfinite(o)).sum()), "LSE nonfinite", int((~torch.isfinite(r[1])).sum()) if lse_flag else "-")
    if lse_flag and (~torch.isfinite(r[1])).any():
        idx = (~torch.isfinite(r[1])).nonzero()[:5]
        print("   bad LSE at", idx.tolist(), "values", r[1][tuple(idx[0].tolist())].item())
    if (~torch.isfinite(o)).any():
        idx = (~torch.isfinite(o)).nonzero()[:5]
        print("   bad O at", idx.tolist())
# structure of the deviation from the oracle on small randint cases
for (S, D) in [(512, 128), (512, 64)]:
    q_, k_, v_ = orc.make_inputs(1, 2, S, D, seed=5, dist="randint")
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).half().to(dev) for x in (q_, k_, v_))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp8_cuda(tq, tk, tv, return_lse=True)
    o_ref, lse_ref = orc.lowbit_fa_forward(q_, k_, v_, return_lse=True, pv="fp8", amax_floor=1e-7)
    on = o.float().cpu().numpy()
    err = np.abs(on - o_ref)
    bad = err > 1e-2 + 2e-2 * np.abs(o_ref)
    rows = bad.any(axis=-1)
    rel = err / (np.abs(o_ref) + 1e-3)
    print(f"S{S} D{D}: bad elements {int(bad.sum())} of {bad.size}; rows with a bad element {int(rows.sum())} of {rows.size}; median row-max relative error on bad rows {np.median(rel[rows].max(axis=-1)):.3f}; max |dLSE| {np.abs(lse.cpu().numpy() - lse_ref).max():.3e}")
    # is the deviation a per-row common factor (weights) ?  ratio O / O_ref on the largest-|ref| channels of bad rows
    r_idx = np.argwhere(rows)[:6]
    for b_, h_, s_ in r_idx:
        ch = np.argsort(-np.abs(o_ref[b_, h_, s_]))[:4]
        print("   row", (int(h_), int(s_)), "O/Oref on its 4 largest channels", np.round(on[b_, h_, s_, ch] / o_ref[b_, h_, s_, ch], 4).tolist())
