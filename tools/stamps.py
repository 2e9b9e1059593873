#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the attention kernel from a -DLBFA_STAMPS build (variants/lib_stamps.so).
   LBFA_LIB_PATH=variants/lib_stamps.so python tools/stamps.py [S] [D]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import _lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B, H = 4, 32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).half() for _ in range(3))
for _ in range(5):
    o = lb.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout="HND", is_causal=False)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(8192 * 8, dtype=np.int64)
lib.lbfa_debug_stamps.argtypes = [ctypes.c_void_p]
rc = lib.lbfa_debug_stamps(buf.ctypes.data)
assert rc == 0, rc
n = B * H * ((S + 127) // 128)
st = buf.reshape(8192, 8)[: min(n, 8192)].astype(np.float64)
t0 = st[:, 0].min()
names = ["prologue(Q load+quant)", "scale table+prime", "tile loop", "vote", "epilogue+store"]
d = np.diff(st[:, :6], axis=1)
tot = st[:, 5] - st[:, 0]
print(f"workgroups {len(st)}  kernel span {(st[:,5].max()-t0):.0f} cycles")
for i, nm in enumerate(names):
    print(f"{nm:26s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}  share {d[:, i].sum() / tot.sum():6.3f}")
print(f"{'total per WG':26s} median {np.median(tot):9.0f}")
if st[:, 6].any():  # in-kernel Q quantiser: 0 -> 6 loads + amax, 6 -> 7 workgroup reduction, 7 -> 1 encode
    for nm, a, b in (("  Q loads + amax", 0, 6), ("  block amax (2 barriers)", 6, 7), ("  encode", 7, 1)):
        print(f"{nm:26s} median {np.median(st[:, b] - st[:, a]):9.0f}")
# start-time histogram: how staggered are the workgroups
start = np.sort(st[:, 0] - t0)
print("start times (cycles) percentiles 10/50/90:", np.percentile(start, [10, 50, 90]).round())
end = st[:, 5] - t0
print("end times percentiles 50/90/99/100:", np.percentile(end, [50, 90, 99, 100]).round())
# first-round vs later rounds
order = np.argsort(st[:, 0])
first, later = order[:768], order[768:]
for nm, idx in (("first 768 WGs", first), ("later WGs", later)):
    if len(idx):
        print(nm, "prologue median", np.median(d[idx, 0]).round(), "loop median", np.median(d[idx, 2]).round(), "total", np.median(tot[idx]).round())
