#!/usr/bin/env python3
"""Operator smoke test - the counterpart of the reference's `example/test_sageattn_operator.py` (:43-104) and
of its bench sweep (`bench/quant/bench_qk_int8_pv_fp16_triton.py`): runs the public API for HND / NHD layouts
with and without the causal flag, prints latency, attention TFLOP/s (4*B*H*D*S*S, halved for causal) and the
MSE against an fp32 SDPA of the un-quantised inputs ("Loss" in the reference's logs).

    python examples/test_operator.py [--kernel int8|int4|q8k4|fp8] [--seq 1024 2048 4096 8192] [--head_dim 64]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lowbit_quant_fa2_paddle_amd as lb


def run_once(fn, q, k, v, layout, causal, repeats):
    for _ in range(5):
        fn(q, k, v, tensor_layout=layout, is_causal=causal)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        o = fn(q, k, v, tensor_layout=layout, is_causal=causal)
    torch.cuda.synchronize()
    return o, (time.perf_counter() - t0) / repeats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="int8", choices=["int8", "int4", "q8k4", "fp8"])
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--num_heads", type=int, default=32)
    ap.add_argument("--head_dim", type=int, default=64)
    ap.add_argument("--seq", type=int, nargs="*", default=[1024, 2048, 4096, 8192])
    ap.add_argument("--repeats", type=int, default=20)
    a = ap.parse_args()
    assert torch.cuda.is_available(), "needs an AMD GPU (gfx950)"
    dev = torch.device("cuda:0")
    fn = {"int8": lb.lowbit_fa_qk_int8_pv_fp16_triton,
          "int4": lb.lowbit_fa_qk_int4_pv_fp16_triton,
          "q8k4": lambda *x, **kw: lb.lowbit_fa_qk_int4_pv_fp16_triton(*x, q_bits=8, **kw),
          "fp8": lb.lowbit_fa_qk_int8_pv_fp8_cuda}[a.kernel]
    B, H, D = a.batch_size, a.num_heads, a.head_dim
    for S in a.seq:
        for layout in ("HND", "NHD"):
            for causal in (False, True):
                shp = (B, H, S, D) if layout == "HND" else (B, S, H, D)
                q, k, v = (torch.randn(shp, device=dev, dtype=torch.float16) for _ in range(3))
                o, dt = run_once(fn, q, k, v, layout, causal, a.repeats)
                flops = 4 * B * H * D * S * S / (2 if causal else 1)
                # accuracy on one (batch, head) slice against exact attention
                sl = (lambda t: t[0, 0]) if layout == "HND" else (lambda t: t[0, :, 0])
                ref = torch.nn.functional.scaled_dot_product_attention(sl(q).float()[None], sl(k).float()[None],
                                                                       sl(v).float()[None], is_causal=causal)[0]
                mse = float(((sl(o).float() - ref) ** 2).mean())
                print(f"{a.kernel} S={S:6d} D={D} {layout} causal={int(causal)}: {flops / dt / 1e12:8.2f} TFLOP/s, "
                      f"{dt * 1e3:8.3f} ms, Loss {mse:.2e}", flush=True)


if __name__ == "__main__":
    main()
