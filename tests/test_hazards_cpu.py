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
