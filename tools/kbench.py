#!/usr/bin/env python3
"""Kernel-only micro-benchmark of lbfa_attn_fwd (pre-quantised operands, as the reference times its kernel,
utils/benchmark.py:240-266) + a quick numerical check against fp32 SDPA.  Development tool.

    python tools/kbench.py [--cfg name ...] [--iters 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import attn_qk_int8_per_block as attn, quant_per_block as qpb, quant

CFG = {  # B,H,S,D,layout,causal,pv
    "c2": (4, 32, 4096, 64, "HND", False, "fp16"),
    "c2c": (4, 32, 4096, 64, "HND", True, "fp16"),
    "s16k": (4, 32, 16384, 64, "HND", False, "fp16"),
    "d128": (4, 32, 4096, 128, "HND", False, "fp16"),
    "s32k": (4, 32, 32768, 64, "HND", False, "fp16"),
    "c5s": (4, 32, 32768, 128, "HND", False, "fp8"),
    "c3": (4, 32, 16384, 128, "NHD", True, "fp16"),
    "c5": (2, 32, 32768, 128, "HND", False, "fp8"),
    "f8d64": (4, 32, 4096, 64, "HND", False, "fp8"),
    # un-quantised kernel (lbfa_sdpa_fwd) and torch's flash SDPA on the same inputs
    "h64": (4, 32, 4096, 64, "HND", False, "qk16"),
    "h64c": (4, 32, 4096, 64, "HND", True, "qk16"),
    "h128": (4, 32, 4096, 128, "HND", False, "qk16"),
    "h128c": (4, 32, 16384, 128, "NHD", True, "qk16"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", nargs="*", default=["c2", "d128", "c2c", "c3", "f8d64"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--dist", default="normal", choices=["normal", "randint"], help="randint: the reference's bench distribution randint(-100, 100)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in a.cfg:
        B, H, S, D, layout, causal, pv = CFG[name]
        shp = (B, H, S, D) if layout == "HND" else (B, S, H, D)
        g = torch.Generator(device=dev); g.manual_seed(0)
        q = torch.randn(shp, generator=g, device=dev).half()
        k = torch.randn(shp, generator=g, device=dev).half()
        v = torch.randn(shp, generator=g, device=dev).half()
        if a.dist == "randint":
            q, k, v = (torch.randint(-100, 100, shp, generator=g, device=dev).half() for _ in range(3))
        if pv == "qk16":
            from lowbit_quant_fa2_paddle_amd import core
            f = lambda: (core.flash_attn_fp16(q, k, v, tensor_layout=layout, is_causal=causal), None)
        else:
            km = qpb.mean_seq(k, layout)
            q8, qs, k8, ks = qpb.per_block_int8(q, k, km=km, sm_scale=D ** -0.5, tensor_layout=layout)
        if pv == "qk16":
            pass
        elif pv == "fp8":
            vin, vs, _ = quant.per_channel_fp8(v, tensor_layout=layout)
        else:
            vin, vs = v, None
        if pv != "qk16":
            f = lambda: attn.forward(q8, k8, vin, qs, ks, tensor_layout=layout, output_dtype=torch.float16, is_causal=causal, v_scale=vs)
        for _ in range(3):
            o, _ = f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(a.iters):
            e0.record(); o, _ = f(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        med = ts[len(ts) // 2]
        fl = 4.0 * B * H * S * S * D / (2 if causal else 1)
        msg = f"{name:6s} med {med:8.4f} ms  min {ts[0]:8.4f} ms  {fl / med / 1e9:8.1f} TFLOP/s (med)  {fl / ts[0] / 1e9:8.1f} (best)"
        if a.check:
            hb = 2
            qq = (q if layout == "HND" else q.transpose(1, 2))[0, :hb].float()
            kk = (k if layout == "HND" else k.transpose(1, 2))[0, :hb].float()
            vv = (v if layout == "HND" else v.transpose(1, 2))[0, :hb].float()
            oo = (o if layout == "HND" else o.transpose(1, 2))[0, :hb].float()
            ref = torch.nn.functional.scaled_dot_product_attention(qq[None], kk[None], vv[None], is_causal=causal)[0]
            msg += f"  mse {float(((oo - ref) ** 2).mean()):.2e} maxabs {float((oo - ref).abs().max()):.2e}"
        if pv == "qk16":  # FA2-class library kernel on the same inputs
            from torch.nn.attention import SDPBackend, sdpa_kernel
            qq, kk, vv = ((x if layout == "HND" else x.transpose(1, 2)) for x in (q, k, v))
            with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
                for _ in range(3):
                    torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, is_causal=causal)
                t2 = []
                for _ in range(a.iters):
                    e0.record(); torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, is_causal=causal); e1.record()
                    torch.cuda.synchronize(); t2.append(e0.elapsed_time(e1))
            t2.sort()
            msg += f"  | torch flash {fl / t2[len(t2) // 2] / 1e9:8.1f} TFLOP/s"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
