#!/usr/bin/env python3
"""Run-to-run bit-stability soak (development aid): every configuration is launched N times and every output compared, bit for bit,
with the first one - a race in the vote / replay machinery of the attention kernels would show as a differing launch.
    python tools/soak.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lowbit_quant_fa2_paddle_amd as lb

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)


def inputs(kind, B, H, S, D, dtype):
    g.manual_seed(1234)
    if kind == "randint":
        q = torch.randint(-100, 100, (B, H, S, D), generator=g, device=dev).to(dtype)
        k = torch.randint(-100, 100, (B, H, S, D), generator=g, device=dev).to(dtype)
    else:
        q = torch.randn((B, H, S, D), generator=g, device=dev)
        k = torch.randn((B, H, S, D), generator=g, device=dev)
        if kind == "exact":      # first key tile far below the rest: every Q block replays on the grid
            a = 8.0 * (D / 64.0) ** 0.5
            q[..., 0] += a; k[:, :, :64, 0] -= a; k[:, :, 64:, 0] += a
        elif kind == "peaky":    # some rows overflow at some vote, others never
            q *= 6.0
        elif kind == "late":     # a 300-binade step in the middle of the sequence: references leave the grid inside replays
            a = 8.0 * (D / 64.0) ** 0.5
            q[..., 0] += a; k[:, :, :64, 0] -= a; k[:, :, 64:S // 2, 0] += a; k[:, :, S // 2:, 0] += 300.0 * D ** 0.5 / (1.44269504 * a)
        q, k = q.to(dtype), k.to(dtype)
    v = torch.randn((B, H, S, D), generator=g, device=dev).to(dtype)
    return q, k, v


bad = 0
for kind in ("normal", "randint", "exact", "peaky", "late"):
    for (B, H, S, D, causal, fn, dtype) in [
        (2, 16, 4096, 64, False, lb.lowbit_fa_qk_int8_pv_fp16_triton, torch.float16),
        (2, 16, 2048, 128, True, lb.lowbit_fa_qk_int8_pv_fp16_triton, torch.float16),
        (2, 8, 1000, 64, True, lb.lowbit_fa_qk_int4_pv_fp16_triton, torch.bfloat16),
        (1, 8, 2048, 128, False, lb.lowbit_fa_qk_int8_pv_fp8_cuda, torch.float16),
    ]:
        q, k, v = inputs(kind, B, H, S, D, dtype)
        o0, l0 = fn(q, k, v, is_causal=causal, return_lse=True)
        ndiff = 0
        for _ in range(N):
            o, l = fn(q, k, v, is_causal=causal, return_lse=True)
            if not (torch.equal(o, o0) and torch.equal(l, l0)):
                ndiff += 1
        ok = torch.isfinite(o0).all().item() and torch.isfinite(l0).all().item()
        print(f"{kind:8s} {fn.__name__[10:]:28s} S{S} D{D} causal={int(causal)} {str(dtype)[6:]}: {ndiff} of {N} launches differ, finite={ok}", flush=True)
        bad += ndiff + (0 if ok else 1)
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
