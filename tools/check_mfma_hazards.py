#!/usr/bin/env python3
"""Static checks of the SHIPPED gfx950 code object: (1) the XDL-write -> VALU-read hazard behind the 16-pass block-scaled MFMA
(below); (2) the hand-counted waits of the inline-asm `ds_read_b64_tr_b16` V^T reads (check_tr_reads); (3) scratch traffic
inside the loops of the attention kernels (scratch_in_loops).

Background (DESIGN.md 3.1, fp8 PV): `v_mfma_scale_f32_32x32x64_f8f6f4` with e4m3 operands takes 16 passes (64 cycles: twice
the cycles of the bf16 32x32 form, MI355X_MICROARCH.md, Matrix cores); with fp6 / fp4 operands the same opcode takes 8.  gfx950
does NOT interlock a VALU / VMEM / LDS / export read of an XDL result: software must leave

        passes + 3 (+ 1 on gfx950 when passes != 2)   = 20 wait states for a 16-pass MFMA before a read,
        passes + 2 (+ 1 on gfx950)                     = 19 before a VALU overwrite (WAW)

(the rule LLVM's GCNHazardRecognizer implements for gfx940 / gfx950, checkMAIVALUHazards).  ROCm 7.2's clang sizes the gap for
the opcode's 8-pass form whatever the operand formats: it leaves 8 + 3 + 1 = 12.  The kernel therefore adds `s_nop 7` (8 wait
states, LBFA_MX_NOP) behind its PV MFMAs: 12 + 8 = 20.  That count is derived, not tuned - and this tool proves it on the
binary: for EVERY such MFMA it walks EVERY successor path (branches followed both ways, loops included) to the first
instruction that touches the destination registers and counts the wait states in between (one per instruction issued,
N + 1 for `s_nop N`; an `s_waitcnt` / `s_barrier` is counted as ONE although it usually lasts far longer).  A touching MFMA
that accumulates onto the same registers (SrcC = vDst chain) is exempt: back-to-back dependent XDL ops are interlocked.

    python tools/check_mfma_hazards.py [path/to/liblowbit_fa_hip.so]      exit code 1 if any path is short
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
NEED_READ = 20   # 32x32x64 with e4m3 operands: 16 passes + 3 + 1
NEED_READ_16 = 12  # 16x16x128 with e4m3 operands: 8 passes + 3 + 1  (not in the shipped library: round 3's 128-key-tile experiment)
HORIZON = 32     # stop following a path after this many wait states


def need_read(mnemonic: str) -> int:
    return NEED_READ_16 if "16x16x128" in mnemonic else NEED_READ

_reg = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_fn = re.compile(r"^([0-9a-f]+) <([^>]+)>:$")
_ins = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_tgt = re.compile(r"<([^>+]+)(?:\+0x([0-9a-f]+))?>\s*$")


def tools_available() -> bool:
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))


def disassemble(so_path: str) -> list[str]:
    """gfx950 disassembly of every code object bundled in the shared library (one bundle per translation unit)."""
    out: list[str] = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(td, "x")], check=True,
                       capture_output=True)
        blob = open(fat, "rb").read()
        pos, i = [], blob.find(MAGIC)
        while i >= 0:
            pos.append(i)
            i = blob.find(MAGIC, i + 1)
        for n, (a, b) in enumerate(zip(pos, pos[1:] + [len(blob)])):
            part, co = os.path.join(td, f"p{n}.bin"), os.path.join(td, f"co{n}.o")
            open(part, "wb").write(blob[a:b])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--targets={TARGET}",
                            f"--output={co}"], check=True, capture_output=True)
            if os.path.getsize(co) == 0:
                continue
            r = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True)
            out += r.stdout.splitlines()
    return out


def _regs(text: str) -> set:
    s = set()
    for m in _reg.finditer(text):
        if m.group(2) is not None:
            s.add((m.group(1), int(m.group(2))))
        else:
            s.update((m.group(1), k) for k in range(int(m.group(3)), int(m.group(4)) + 1))
    return s


def parse(lines: list[str]) -> dict:
    """name -> list of (addr, mnemonic, operand text, branch target addr or None)"""
    fns, cur, base = {}, None, {}
    for ln in lines:
        m = _fn.match(ln)
        if m:
            cur = m.group(2)
            base[cur] = int(m.group(1), 16)
            fns[cur] = []
            continue
        m = _ins.match(ln)
        if m and cur is not None:
            mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
            tgt = None
            if mn.startswith("s_cbranch") or mn == "s_branch":
                t = _tgt.search(ln)
                if t and t.group(1) in base:
                    tgt = base[t.group(1)] + (int(t.group(2), 16) if t.group(2) else 0)
            fns[cur].append((addr, mn, ops, tgt))
    return fns


def check_function(ins: list) -> list:
    """[(mfma addr, min wait states to the first non-exempt touch of its result, touching instruction)] for every scaled MFMA"""
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    res = []
    for i, (addr, mn, ops, _) in enumerate(ins):
        if "mfma_scale" not in mn:
            continue
        dst = _regs(ops.split(",")[0])
        best, best_ins = HORIZON, None
        seen: dict = {}
        stack = [(i + 1, 0)]
        while stack:
            j, w = stack.pop()
            while j < len(ins):
                if w >= HORIZON or seen.get(j, HORIZON + 1) <= w:
                    break
                seen[j] = w
                a2, mn2, ops2, tgt2 = ins[j]
                if _regs(ops2) & dst:
                    if mn2.startswith("v_mfma") and _regs(ops2.split(",")[0]) == _regs(ops2.split(",")[3] if ops2.count(",") >= 3 else ""):
                        pass  # accumulate chain onto the same registers: interlocked; that MFMA is checked on its own
                    elif w < best:
                        best, best_ins = w, f"{a2:x}: {mn2} {ops2}"
                    break
                if mn2 == "s_endpgm" or mn2.startswith("s_setpc") or mn2.startswith("s_swappc"):
                    break
                w += int(ops2.split()[0], 0) + 1 if mn2 == "s_nop" else 1
                if mn2 == "s_branch":
                    if tgt2 in index:
                        j = index[tgt2]
                        continue
                    break
                if mn2.startswith("s_cbranch") and tgt2 in index:
                    stack.append((index[tgt2], w))
                j += 1
        res.append((addr, best, best_ins, need_read(mn)))
    return res


_LGKM_PREFIXES = ("ds_", "s_load", "s_buffer_load", "s_scratch_load", "s_dcache", "s_memtime", "s_memrealtime", "s_sendmsg",
                  "s_atc_probe", "s_atomic", "s_buffer_atomic", "s_store", "s_buffer_store")
_lgkm = re.compile(r"lgkmcnt\((\d+)\)")


def check_tr_reads(ins: list) -> list:
    """The V^T fragments are read by inline-asm `ds_read_b64_tr_b16` whose completion the source counts by hand
    (attn_common.h, lds_wait_keep).  For every such read, on every successor path: no instruction may read or write a destination
    register before an `s_waitcnt lgkmcnt(n)` that covers the read - n <= the number of LDS operations issued after it (LDS
    returns in order), or n == 0 once a scalar-memory operation (out of order) has been issued in between.
    -> [(addr, offending instruction)]"""
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    bad = []
    for i, (addr, mn, ops, _) in enumerate(ins):
        if mn != "ds_read_b64_tr_b16":
            continue
        dst = _regs(ops.split(",")[0])
        seen: dict = {}
        stack = [(i + 1, 0, False)]
        hit = None
        while stack and hit is None:
            j, younger, smem = stack.pop()
            while j < len(ins):
                key = (younger, smem)
                if key in seen.setdefault(j, set()):
                    break
                seen[j].add(key)
                a2, mn2, ops2, tgt2 = ins[j]
                if mn2 == "s_waitcnt":
                    m = _lgkm.search(ops2)
                    if m is not None:
                        n = int(m.group(1))
                        if n == 0 or (not smem and n <= younger):
                            break  # covered on this path
                elif _regs(ops2) & dst:
                    hit = f"{a2:x}: {mn2} {ops2}"
                    break
                if mn2 == "s_endpgm" or mn2.startswith("s_setpc") or mn2.startswith("s_swappc"):
                    hit = f"{a2:x}: {mn2} (read never waited for)"
                    break
                if mn2.startswith(_LGKM_PREFIXES):
                    if mn2.startswith("ds_"):
                        younger += 1
                    else:
                        smem = True
                    younger = min(younger, 64)
                if mn2 == "s_branch":
                    if tgt2 in index:
                        j = index[tgt2]
                        continue
                    break
                if mn2.startswith("s_cbranch") and tgt2 in index:
                    stack.append((index[tgt2], younger, smem))
                j += 1
        if hit is not None:
            bad.append((addr, hit))
    return bad


def scratch_in_loops(ins: list) -> int:
    """number of scratch_* instructions inside the INNERMOST loops of one function (address ranges closed by a backward branch
    that contain no other such range: the unrolled tile loops of the attention kernels, not the replay around them)"""
    spans = sorted({(tgt, a) for (a, mn, _, tgt) in ins if (mn.startswith("s_cbranch") or mn == "s_branch") and tgt is not None and tgt <= a})
    inner = [(lo, hi) for (lo, hi) in spans if not any((lo2, hi2) != (lo, hi) and lo <= lo2 and hi2 <= hi for (lo2, hi2) in spans)]
    n = 0
    for a, mn, _, _ in ins:
        if mn.startswith("scratch_") and any(lo <= a <= hi for lo, hi in inner):
            n += 1
    return n


def check_attention_kernels(so_path: str) -> dict:
    """tr-read waits and in-loop scratch traffic of every attention kernel instance in the library"""
    fns = parse(disassemble(so_path))
    rep = {"tr_reads": 0, "tr_violations": [], "scratch_in_loops": {}}
    for name, ins in fns.items():
        if "attn_fwd" not in name:
            continue
        rep["tr_reads"] += sum(1 for x in ins if x[1] == "ds_read_b64_tr_b16")
        for addr, what in check_tr_reads(ins):
            rep["tr_violations"].append((name, hex(addr), what))
        n = scratch_in_loops(ins)
        if n:
            rep["scratch_in_loops"][name] = n
    return rep


def check(so_path: str) -> dict:
    fns = parse(disassemble(so_path))
    report = {"mfma_scale": 0, "min_wait_states": HORIZON, "short": [], "required": 0}
    for name, ins in fns.items():
        for addr, w, what, need in check_function(ins):
            report["mfma_scale"] += 1
            report["required"] = max(report["required"], need)
            if w < report["min_wait_states"]:
                report["min_wait_states"], report["closest"] = w, f"{name} @{addr:x} -> {what}"
            if w < need:
                report["short"].append((name, hex(addr), w, what))
    return report


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "lowbit_quant_fa2_paddle_amd", "liblowbit_fa_hip.so")
    rep = check(so)
    print(f"{rep['mfma_scale']} block-scaled MFMAs; fewest wait states before a dependent non-MFMA access: {rep['min_wait_states']}"
          f"{'+' if rep['min_wait_states'] >= HORIZON else ''} (required {rep['required']})")
    if "closest" in rep and rep["min_wait_states"] < HORIZON:
        print("closest:", rep["closest"])
    for s in rep["short"][:20]:
        print("SHORT:", s)
    rep2 = check_attention_kernels(so)
    print(f"{rep2['tr_reads']} hand-issued ds_read_b64_tr_b16; destination touched before a covering lgkmcnt wait: {len(rep2['tr_violations'])}")
    for s in rep2["tr_violations"][:20]:
        print("EARLY:", s)
    print(f"attention kernels with scratch traffic inside a loop: {len(rep2['scratch_in_loops'])}")
    for k, v in sorted(rep2["scratch_in_loops"].items()):
        print(f"  {v:4d}  {k}")
    sys.exit(1 if (rep["short"] or rep2["tr_violations"]) else 0)
