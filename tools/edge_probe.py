#!/usr/bin/env python3
"""Development probe: adversarial input distributions through every operator - finiteness, and distance from the oracle in units of
the test tolerance (2e-3 + 2e-3 |O| scaled by max|v| for the fp16-P operators, 1e-2 + 2e-2 |O| for fp8 PV)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import core
from oracle import lowbit_fa_oracle as orc
dev = torch.device("cuda:0")
S, H = 640, 2


def dists(D, rng):
    n = lambda *s: rng.standard_normal(s).astype(np.float32)
    base = lambda: (n(1, H, S, D), n(1, H, S, D), n(1, H, S, D))
    out = {}
    q, k, v = base(); out["huge qk x1000"] = (q * 1000, k * 1000, v)
    q, k, v = base(); out["huge qk x60"] = (q * 60, k * 60, v)
    q, k, v = base(); out["tiny qk x1e-4"] = (q * 1e-4, k * 1e-4, v)
    q, k, v = base(); out["q = 0"] = (q * 0, k, v)
    q, k, v = base(); out["k = 0"] = (q, k * 0, v)
    q, k, v = base(); out["k constant"] = (q, np.ones_like(k) * 3.0, v)
    q, k, v = base(); m = rng.random(k.shape) < 0.01; out["k sparse spikes"] = (q, np.where(m, 500.0, 0.0).astype(np.float32) * np.sign(k), v)
    q, k, v = base(); sc = np.where((np.arange(S) // 64) % 2 == 0, 1000.0, 1e-3).astype(np.float32); out["k blocks x1000 / x1e-3"] = (q, k * sc[None, None, :, None], v)
    q, k, v = base(); out["v x10000"] = (q, k, v * 10000)
    q, k, v = base(); out["v = 0"] = (q, k, v * 0)
    q, k, v = base(); q[..., 0] = 60000.0; k[..., 0] = 1.0; out["q channel at fp16 max"] = (q, k, v)
    q, k, v = base(); out["one query block huge"] = (np.where((np.arange(S) // 128 == 2)[None, None, :, None], q * 2000, q), k, v)
    return out


for D in (64, 128):
    rng = np.random.default_rng(D)
    for name, (q, k, v) in dists(D, rng).items():
        q, k, v = (orc.to_storage(np.clip(x, -65000, 65000), "fp16") for x in (q, k, v))
        tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).half().to(dev) for x in (q, k, v))
        vmax = max(float(np.abs(v).max()), 1.0)
        for causal in (False, True):
            line = f"D{D} {name:26s} causal={int(causal)}:"
            for op, fn, kw, okw in (("int8", lb.lowbit_fa_qk_int8_pv_fp16_triton, {}, {}),
                                    ("int4", lb.lowbit_fa_qk_int4_pv_fp16_triton, {}, dict(q_qmax=7, k_qmax=7)),
                                    ("fp8", lb.lowbit_fa_qk_int8_pv_fp8_cuda, {}, dict(pv="fp8"))):
                o, lse = fn(tq, tk, tv, is_causal=causal, return_lse=True, **kw)
                fin = bool(torch.isfinite(o).all() and torch.isfinite(lse).all())
                with np.errstate(all="ignore"):
                    tail = {} if op == "fp8" else dict(tail="neg_inf")
                    o_ref, lse_ref = orc.lowbit_fa_forward(q, k, v, is_causal=causal, return_lse=True, amax_floor=1e-7, **tail, **okw)
                on = o.float().cpu().numpy()
                rfin = bool(np.isfinite(o_ref).all())
                if fin and rfin:
                    err = np.abs(on - o_ref)
                    tol = (1e-2 * vmax + 2e-2 * np.abs(o_ref)) if op == "fp8" else (2e-3 * vmax + 2e-3 * np.abs(o_ref))
                    line += f"  {op} {np.max(err / tol):6.2f}x tol"
                else:
                    line += f"  {op} finite={fin} (oracle finite={rfin})"
            of = core.flash_attn_fp16(tq, tk, tv, is_causal=causal)
            ref = orc.sdpa_naive(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), is_causal=causal)
            f2 = bool(torch.isfinite(of).all())
            e2 = np.abs(of.float().cpu().numpy() - ref)
            line += f"  sdpa16 {'%6.2fx tol' % np.max(e2 / (2e-3 * vmax + 2e-3 * np.abs(ref))) if f2 else 'finite=False'}"
            print(line, flush=True)
