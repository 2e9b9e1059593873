#!/usr/bin/env python3
"""Development probe: q, k of very large magnitude (fp16 and bf16 storage) through the quantised operators, in detail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowbit_quant_fa2_paddle_amd as lb
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
S, H = 640, 2
for D in (64, 128):
    for dt, mul in (("fp16", 1000.0), ("fp16", 1e4), ("bf16", 1e4), ("bf16", 1e5), ("bf16", 1e7)):
        rng = np.random.default_rng(D)
        q, k, v = (rng.standard_normal((1, H, S, D)).astype(np.float32) for _ in range(3))
        q, k = orc.to_storage(q * mul, dt), orc.to_storage(k * mul, dt)
        v = orc.to_storage(v, dt)
        tdt = torch.bfloat16 if dt == "bf16" else torch.float16
        tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to(tdt).to(dev) for x in (q, k, v))
        for smooth in (True, False):
            for op, fn, okw in (("int8", lb.lowbit_fa_qk_int8_pv_fp16_triton, {}), ("fp8", lb.lowbit_fa_qk_int8_pv_fp8_cuda, dict(pv="fp8"))):
                o, lse = fn(tq, tk, tv, return_lse=True, smooth_k=smooth)
                with np.errstate(all="ignore"):
                    o_ref, lse_ref = orc.lowbit_fa_forward(q, k, v, return_lse=True, smooth_k=smooth, amax_floor=1e-7, dtype=dt, **okw)
                on, ln = o.float().cpu().numpy(), lse.cpu().numpy()
                bad_rows = (~np.isfinite(on)).any(-1)
                msg = f"D{D} {dt} x{mul:g} smooth={int(smooth)} {op}: O nonfinite rows {int(bad_rows.sum())}/{bad_rows.size}, LSE nonfinite {int((~np.isfinite(ln)).sum())} (oracle O nonfinite {int((~np.isfinite(o_ref)).sum())}, LSE {int((~np.isfinite(lse_ref)).sum())})"
                ok = np.isfinite(on).all(-1) & np.isfinite(o_ref).all(-1)
                if ok.any():
                    err = np.abs(on - o_ref)[ok]
                    rel = (2e-3 + (2.0 ** -7 if dt == "bf16" else 0))
                    tol = (1e-2 + 2e-2 * np.abs(o_ref[ok])) if op == "fp8" else (2e-3 + rel * np.abs(o_ref[ok]))
                    rows_bad = (err > tol).any(-1)
                    msg += f"; finite rows: worst err/tol {np.max(err / tol):.2f}, rows beyond tol {int(rows_bad.sum())}/{rows_bad.size}"
                print(msg, flush=True)
