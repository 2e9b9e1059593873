#!/usr/bin/env python3
"""Development probe: very long sequences (S = 131072 and a ragged 100001) - indexing sanity against fp32 attention on sampled rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import core
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
for S, D, layout in ((131072, 128, "HND"), (100001, 64, "NHD"), (262144, 64, "HND")):
    B, H = 1, 2
    shp = (B, H, S, D) if layout == "HND" else (B, S, H, D)
    q, k, v = (torch.randn(shp, generator=g, device=dev).half() for _ in range(3))
    rows = torch.tensor([0, 1, 63, 64, 127, 128, 4095, 65535, 65536, S // 2, S - 129, S - 2, S - 1], device=dev)
    can = (lambda t: t) if layout == "HND" else (lambda t: t.transpose(1, 2))
    for causal in (False, True):
        for name, fn in (("int8", lb.lowbit_fa_qk_int8_pv_fp16_triton), ("fp8", lb.lowbit_fa_qk_int8_pv_fp8_cuda), ("sdpa16", None)):
            if fn is None:
                o, lse = core.flash_attn_fp16(q, k, v, tensor_layout=layout, is_causal=causal, return_lse=True)
            else:
                o, lse = fn(q, k, v, tensor_layout=layout, is_causal=causal, return_lse=True)
            torch.cuda.synchronize()
            qq, kk, vv, oo = can(q)[0].float(), can(k)[0].float(), can(v)[0].float(), can(o)[0].float()
            s = torch.einsum("hrd,hkd->hrk", qq[:, rows], kk) * D ** -0.5
            if causal:
                s = s.masked_fill(torch.arange(S, device=dev)[None, None, :] > rows[None, :, None], float("-inf"))
            ref = torch.einsum("hrk,hkd->hrd", torch.softmax(s, -1), vv)
            lref = torch.logsumexp(s, -1)
            err = (oo[:, rows] - ref).abs().max().item()
            lerr = (lse[0][:, rows] - lref).abs().max().item()
            print(f"S{S} D{D} {layout} causal={int(causal)} {name}: finite {bool(torch.isfinite(o).all())} max|dO| on sampled rows {err:.2e} (|ref| max {ref.abs().max().item():.2e}) max|dLSE| {lerr:.2e}", flush=True)
    del q, k, v
