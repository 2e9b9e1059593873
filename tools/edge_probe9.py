#!/usr/bin/env python3
"""Development probe: the one fp8-PV element of edge_probe7's fused-qkv case beyond the per-element bound - which row, how many keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowbit_quant_fa2_paddle_amd as lb
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
B, S, H, D = 2, 333, 4, 64
qkv = torch.randn((B, S, 3, H, D), generator=g, device=dev).half()
tq, tk, tv = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
o, lse = lb.lowbit_fa_qk_int8_pv_fp8_cuda(tq, tk, tv, tensor_layout="NHD", is_causal=True, return_lse=True)
o_ref, lse_ref = orc.lowbit_fa_forward(q, k, v, tensor_layout="NHD", is_causal=True, return_lse=True, amax_floor=1e-7, pv="fp8")
on = o.float().cpu().numpy()
err = np.abs(on - o_ref); tol = 1e-2 + 2e-2 * np.abs(o_ref)
idx = np.argwhere(err > tol)
print("elements beyond the bound:", len(idx), "of", err.size)
for i in idx[:10]:
    b, s, h, d = i
    print(f"  batch {b} token {s} (sees {s + 1} keys) head {h} channel {d}: got {on[b, s, h, d]:.5f} ref {o_ref[b, s, h, d]:.5f} err {err[b, s, h, d]:.4f} tol {tol[b, s, h, d]:.4f}")
r = err / tol
print("rows (token index) with err/tol > 0.7:", sorted(set(np.argwhere(r > 0.7)[:, 1].tolist()))[:40])
