#!/usr/bin/env python3
"""The reference's headline sweep (bench/quant/bench_qk_int8_pv_fp16_triton.py): B4 H32, S in {4096, 8192, 16384,
32768} x D in {64, 128} x {non-causal, causal}, N(0,1) fp16 inputs - whole operator and attention kernel only
(median of --iters launches, HIP events), next to torch's flash SDPA and this library's un-quantised kernel on the same
inputs.  Writes profiles/<tag>_headline.json and prints a markdown table.   python tools/headline_table.py [--tag r01]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import _lib, core


def med_ms(f, iters):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def kernel_ms(f, iters):
    """attention launch only, bracketed by events the library records itself (lbfa_profile_next_attn)"""
    lib = _lib.load()
    ts = []
    for _ in range(iters + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        lib.lbfa_profile_next_attn(e0.cuda_event, e1.cuda_event)
        f(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--api", default="int8_fp16", choices=["int8_fp16", "int8_fp8", "int4_fp16"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    fn = {"int8_fp16": lb.lowbit_fa_qk_int8_pv_fp16_triton, "int8_fp8": lb.lowbit_fa_qk_int8_pv_fp8_cuda,
          "int4_fp16": lb.lowbit_fa_qk_int4_pv_fp16_triton}[a.api]
    from torch.nn.attention import SDPBackend, sdpa_kernel
    B, H = 4, 32
    rows = []
    print(f"| S | D | causal | whole op TFLOP/s | kernel TFLOP/s | ms (whole) | torch flash SDPA | own fp16 kernel | whole / torch |")
    print("|---|---|---|---|---|---|---|---|---|")
    for D in (64, 128):
        for causal in (False, True):
            for S in (4096, 8192, 16384, 32768):
                g = torch.Generator(device=dev); g.manual_seed(S + D)
                q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).half() for _ in range(3))
                fl = 4.0 * B * H * S * S * D / (2 if causal else 1)
                it = max(3, a.iters if S <= 8192 else a.iters // 2)
                t_op = med_ms(lambda: fn(q, k, v, is_causal=causal), it)
                t_k = kernel_ms(lambda: fn(q, k, v, is_causal=causal), it)
                t_own = med_ms(lambda: core.flash_attn_fp16(q, k, v, is_causal=causal), it)
                with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
                    t_fa = med_ms(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal), it)
                r = dict(api=a.api, B=B, H=H, S=S, D=D, causal=causal, whole_tflops=fl / t_op / 1e9, kernel_tflops=fl / t_k / 1e9,
                         whole_ms=t_op, kernel_ms=t_k, torch_flash_tflops=fl / t_fa / 1e9, own_fp16_tflops=fl / t_own / 1e9)
                rows.append(r)
                print(f"| {S} | {D} | {int(causal)} | {r['whole_tflops']:.0f} | {r['kernel_tflops']:.0f} | {t_op:.3f} | "
                      f"{r['torch_flash_tflops']:.0f} | {r['own_fp16_tflops']:.0f} | {t_fa / t_op:.2f}x |", flush=True)
                del q, k, v
    os.makedirs("profiles", exist_ok=True)
    json.dump(rows, open(f"profiles/{a.tag}_headline_{a.api}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
