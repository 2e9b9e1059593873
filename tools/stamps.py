#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the attention kernel from a -DLBFA_STAMPS16 build of attn_fwd16.hip (tools/build_exp.sh stamps16 "-DLBFA_STAMPS16").
   LBFA_LIB_PATH=variants/lib_stamps128.so python tools/stamps.py [S] [D] [causal]   (build: tools/build_exp.sh stamps128 "-DLBFA_STAMPS16=128")"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowbit_quant_fa2_paddle_amd as lb
from lowbit_quant_fa2_paddle_amd import _lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B, H = 4, 32
CAUSAL = len(sys.argv) > 3 and sys.argv[3] == "causal"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = (torch.randn((B, H, S, D), generator=g, device=dev).half() for _ in range(3))
import time
t0 = time.time()
while time.time() - t0 < 2.0:  # >= 2 s of back-to-back launches: the clock has settled under load
    for _ in range(4):
        o = lb.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout="HND", is_causal=CAUSAL)
    torch.cuda.synchronize()
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(8192 * 24, dtype=np.int64)
lib.lbfa_debug_stamps.argtypes = [ctypes.c_void_p]
rc = lib.lbfa_debug_stamps(buf.ctypes.data)
assert rc == 0, rc
n = B * H * ((S + 127) // 128)
rec = buf.reshape(8192, 24)[: min(n, 8192)].astype(np.float64)
st = rec[:, :8]
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
# in-kernel clock over the tile loop: d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
rt = rec[:, 17] - rec[:, 16]
okc = rt > 0
if okc.any():
    clk = (st[okc, 3] - st[okc, 2]) / rt[okc] * 0.1
    print(f"in-kernel clock over the tile loop: median {np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}); "
          f"tile loop median {np.median(rt[okc]) * 10:.0f} ns")
# one tile of the lazy main loop, wave 0: progress of the instruction stream (s_memtime ticks = shader clocks)
tt = rec[:, 8:16]
ok = (tt > 0).all(axis=1)
if ok.any():
    dd = np.diff(tt[ok], axis=1)
    lab = ["fetch issue (DMA)", "QK^T issue", "exp k-step 0", "PV k-step 0 + exp k-step 1", "PV k-step 1", "wait vmcnt(0)", "barrier"]
    tot_t = tt[ok, 7] - tt[ok, 0]
    print(f"one tile, wave 0 ({ok.sum()} workgroups): total median {np.median(tot_t):7.0f} ticks")
    for i, nm in enumerate(lab):
        print(f"   {nm:30s} median {np.median(dd[:, i]):7.0f}  mean {dd[:, i].mean():7.1f}  share {dd[:, i].sum() / tot_t.sum():6.3f}")
# The counter is per XCD (blockIdx % 8, unsynchronised): slot utilisation and the idle time between a workgroup's end and its
# successor's start are computed within each XCD.
slots = 32 * (3 if D == 64 else 2)
for x in range(8):
    idx = np.arange(x, len(st), 8)
    s0, s5 = st[idx, 0], st[idx, 5]
    span = s5.max() - s0.min()
    busy = (s5 - s0).sum() / slots
    # greedy slot reconstruction: each start takes over the slot that ended last before it
    ends = np.sort(s5)
    gaps = []
    for t_start in np.sort(s0)[slots:]:
        k = np.searchsorted(ends, t_start) - 1
        if k >= 0:
            gaps.append(t_start - ends[k])
    if x < 2 or x == 7:
        print(f"XCD {x}: span {span:9.0f} cycles, slots busy {busy / span:6.3f}, start-after-end gap median {np.median(gaps) if gaps else 0:7.0f} "
              f"(min over predecessors), first-round start spread {np.sort(s0)[:slots].max() - s0.min():7.0f}")
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
