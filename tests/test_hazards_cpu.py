"""CPU check of the SHIPPED gfx950 code object (no GPU needed): every block-scaled fp8 MFMA of the attention kernel is
followed, on every control-flow path, by at least the 20 wait states gfx950 requires before a non-MFMA access to its
result (tools/check_mfma_hazards.py explains the count).  Guards a future recompile - another compiler version, -O level
or kernel edit - against re-opening the stale-accumulator bug round 1 worked around with `s_nop 7` (LBFA_MX_NOP)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_mfma_hazards as chk  # noqa: E402


@pytest.mark.skipif(not chk.tools_available(), reason="llvm-objdump / clang-offload-bundler not found under /opt/rocm")
def test_scaled_mfma_results_are_not_read_early():
    so = os.path.join(ROOT, "lowbit_quant_fa2_paddle_amd", "liblowbit_fa_hip.so")
    assert os.path.exists(so), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    rep = chk.check(so)
    assert rep["mfma_scale"] >= 32, rep  # the fp8-PV instances are in the binary
    assert not rep["short"], rep["short"][:5]
    assert rep["min_wait_states"] >= rep["required"]


@pytest.fixture(scope="module")
def attention_report():
    so = os.path.join(ROOT, "lowbit_quant_fa2_paddle_amd", "liblowbit_fa_hip.so")
    assert os.path.exists(so), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    return chk.check_attention_kernels(so)


@pytest.mark.skipif(not chk.tools_available(), reason="llvm-objdump / clang-offload-bundler not found under /opt/rocm")
def test_hand_issued_lds_transpose_reads_are_waited_for(attention_report):
    """The V^T fragments are read by inline-asm ds_read_b64_tr_b16 whose completion the SOURCE counts (attn_common.h,
    lds_wait_keep): the register allocator is free to copy or spill such a destination between the read and the wait, and the
    hardware would then use stale V without a word.  Every such read of the shipped object, every successor path: no instruction
    touches a destination before an `s_waitcnt lgkmcnt(n)` that covers the read."""
    assert attention_report["tr_reads"] >= 1000, attention_report["tr_reads"]  # the fp16-P kernels are in the binary
    assert not attention_report["tr_violations"], attention_report["tr_violations"][:5]


# in-loop scratch traffic the build consciously keeps (instance -> instructions inside its innermost loops); everything else: 0
SCRATCH_ALLOWED = {
    # un-quantised causal D = 128 instances at the register limit of their occupancy (256): a handful of reloads per two tiles.
    "attn_fwd16_kernelILi128ELi0ELi0ELi0ELb1ELb0E": 8,
    "attn_fwd16_kernelILi128ELi1ELi1ELi1ELb1ELb0E": 8,
    # causal int8 D = 64 instances with the in-kernel Q quantiser: a few reloads (one dword pair, one dword) in the loop over the masked diagonal tiles, which runs
    # ONCE per Q block (tools/scratch_report.py names the loop; the main tile loops, lazy and replay, are clean - round 3 carried 6
    # reloads in them)
    "attn_fwd16_kernelILi64ELi3ELi0ELi0ELb1ELb1E": 4,
    "attn_fwd16_kernelILi64ELi3ELi0ELi1ELb1ELb1E": 4,
}


@pytest.mark.skipif(not chk.tools_available(), reason="llvm-objdump / clang-offload-bundler not found under /opt/rocm")
def test_no_scratch_traffic_inside_the_tile_loops(attention_report):
    """Spills inside the tile loops cost an issue-bound kernel twice (the instruction and its HBM write-back): fail when an
    attention instance that is not on the allow-list above grows any, or a listed one grows beyond what was accepted."""
    over = {}
    for name, n in attention_report["scratch_in_loops"].items():
        allowed = max([v for k, v in SCRATCH_ALLOWED.items() if k in name] or [0])
        if n > allowed:
            over[name] = (n, allowed)
    assert not over, over
