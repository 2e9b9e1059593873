#!/usr/bin/env python3
"""Development probe: strided views (q, k, v sliced out of one fused qkv tensor; head slices; last-dim slices) and unusual sm_scale
values through the operators, against the oracle on the same values."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import core
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
OPS = (("int8", lb.lowbit_fa_qk_int8_pv_fp16_triton, dict(tail="neg_inf"), (2e-3, 2e-3)), ("fp8", lb.lowbit_fa_qk_int8_pv_fp8_cuda, dict(pv="fp8"), (1e-2, 2e-2)))

def check(tag, tq, tk, tv, layout, causal, sm_scale=None):
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    for op, fn, okw, (at, rt) in OPS:
        try:
            kw = {} if sm_scale is None else dict(sm_scale=sm_scale)
            o, lse = fn(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True, **kw)
        except Exception as e:  # noqa: BLE001
            print(f"{tag} {op}: raised {type(e).__name__}: {str(e)[:150]}", flush=True)
            continue
        with np.errstate(all="ignore"):
            o_ref, lse_ref = orc.lowbit_fa_forward(q, k, v, tensor_layout=layout, is_causal=causal, return_lse=True, amax_floor=1e-7, sm_scale=sm_scale, **okw)
        on, ln = o.float().cpu().numpy(), lse.cpu().numpy()
        err = np.abs(on - o_ref) / (at + rt * np.abs(o_ref))
        print(f"{tag} {op}: finite {bool(np.isfinite(on).all() and np.isfinite(ln).all())} worst err/tol {np.nanmax(err):.2f} max|dLSE| {np.nanmax(np.abs(ln - lse_ref)):.2e} contiguous out {o.is_contiguous()} shape {tuple(o.shape)}", flush=True)

g = torch.Generator(device=dev); g.manual_seed(7)
B, S, H, D = 2, 333, 4, 64
qkv = torch.randn((B, S, 3, H, D), generator=g, device=dev).half()
check("fused qkv NHD views", qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], "NHD", True)
qkv2 = torch.randn((B, 3, H, S, 128), generator=g, device=dev).half()
check("fused qkv HND views D128", qkv2[:, 0], qkv2[:, 1], qkv2[:, 2], "HND", False)
big = torch.randn((B, 8, 400, 128), generator=g, device=dev).half()
check("head + seq + channel slices (D 128 -> 64 view)", big[:, 2:6, 10:330, :64], big[:, 4:6, 20:340, 64:], big[:, 0:2, 5:325, 32:96], "HND", True)
check("every second token", big[:, :4, ::2, :], big[:, 4:, ::2, :], big[:, :4, 1::2, :], "HND", False)
tr = torch.randn((B, 300, 4, 64), generator=g, device=dev).half()
check("NHD tensor passed as HND via transpose", tr.transpose(1, 2), tr.transpose(1, 2), tr.transpose(1, 2), "HND", True)
x = torch.randn((1, 2, 256, 64), generator=g, device=dev).half()
for sms in (-0.125, 1e-6, 10.0, 1.0):
    check(f"sm_scale {sms}", x, x.flip(2), x.roll(3, 2), "HND", False, sm_scale=sms)
# un-quantised kernel on the fused views
o = core.flash_attn_fp16(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], tensor_layout="NHD", is_causal=True)
ref = orc.sdpa_naive(*(t.permute(0, 2, 1, 3).float().cpu().numpy().astype(np.float64) for t in (qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])), is_causal=True)
print("sdpa16 fused views: worst err/tol", float(np.max(np.abs(o.permute(0, 2, 1, 3).float().cpu().numpy() - ref) / (2e-3 + 2e-3 * np.abs(ref)))), flush=True)
