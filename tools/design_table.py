#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 3.1 from a bench.py JSON line (default profiles/r03_bench_line.json)."""
import json, sys
d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_bench_line.json").read().strip().splitlines()[-1])
f = d["fa2_reference"]
print("| workload (B4 H32, N(0,1) fp16 inputs) | whole op | attention kernel | frac of roof | torch FA2 | own fp16 | whole / torch FA2 | whole / own fp16 |")
print("|---|---|---|---|---|---|---|---|")
print(f"| **C2 = the bench `value`** (int8/fp16 S4096 D64, {d['steps']} steps) | **{d['value']:.0f}** | **{d['roofline']['achieved']:.0f}** | **{d['roofline']['frac']:.3f}** | "
      f"{f['tflops']:.0f} | {f['own_fp16_kernel']['tflops']:.0f} | {f['speedup_whole_op']:.2f}× | {f['own_fp16_kernel']['lowbit_speedup_whole_op']:.2f}× |")
for r in d["sweep"]:
    g = lambda k, fmt="{:.0f}": (fmt.format(r[k]) if r.get(k) is not None else "—")
    print(f"| {r['workload']} | {g('tflops')} | {g('kernel_tflops')} | {r['frac']:.3f} | {g('torch_fa2_tflops')} | {g('own_fp16_tflops')} | "
          f"{g('vs_torch_fa2', '{:.2f}×')} | {g('vs_own_fp16', '{:.2f}×')} |")
c = d.get("c5_strong")
if c:
    print(f"| C5 itself: B=32 on {d['n_gpus']} GPU(s) (`c5_strong`) | {c['tflops_total']:.0f} ({c['ms']:.0f} ms) | | | | | | |")
