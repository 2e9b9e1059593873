#!/usr/bin/env python3
"""In-loop scratch instructions per attention-kernel instance of the built library (development aid; the CPU suite asserts the same)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_mfma_hazards as chk
from lowbit_quant_fa2_paddle_amd import _lib
rep = chk.check_attention_kernels(_lib.LIB_PATH)
for k, v in rep["scratch_in_loops"].items():
    if v:
        print(k.split("attn_fwd")[1][:60], v)
print("tr reads", rep["tr_reads"], "violations", len(rep["tr_violations"]))
# which loops: every innermost backward-branch span with its size, MFMA count and scratch instructions
fns = chk.parse(chk.disassemble(_lib.LIB_PATH))
for name, ins in fns.items():
    if name not in rep["scratch_in_loops"]:
        continue
    spans = sorted({(tgt, a) for (a, mn, _, tgt) in ins if (mn.startswith("s_cbranch") or mn == "s_branch") and tgt is not None and tgt <= a})
    inner = [(lo, hi) for (lo, hi) in spans if not any((lo2, hi2) != (lo, hi) and lo <= lo2 and hi2 <= hi for (lo2, hi2) in spans)]
    base = ins[0][0]
    print(name.split("attn_fwd")[1][:48])
    for lo, hi in inner:
        nm = sum(1 for a, mn, _, _ in ins if lo <= a <= hi and mn.startswith("v_mfma"))
        ns = [mn for a, mn, _, _ in ins if lo <= a <= hi and mn.startswith("scratch_")]
        if nm:
            print(f"    loop at +{lo - base:#x}: {hi - lo} bytes, {nm} MFMAs, scratch {ns}")
