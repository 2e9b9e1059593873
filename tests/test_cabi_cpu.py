"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/lowbit_fa.h declares, and
rejects bad arguments with the reference's messages before touching the GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "lowbit_fa.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lbfa_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from lowbit_quant_fa2_paddle_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from lowbit_quant_fa2_paddle_amd import _lib
    declared = _declared_symbols()
    assert declared, "no declarations parsed from the header"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/lowbit_fa.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes SIGNATURES out of sync with the header"
    assert lib.lbfa_version() == 200


def test_sizes_are_pure_functions(lib):
    # [B,H,nsplit,D] fp64 partial sums with 256-row splits up to 16K keys
    assert lib.lbfa_mean_seq_workspace_bytes(4, 32, 4096, 64) == 4 * 32 * 16 * 64 * 8
    assert lib.lbfa_mean_seq_workspace_bytes(0, 32, 4096, 64) == 0
    assert lib.lbfa_v_fp8_bytes(1, 2, 100, 64) == 1 * 2 * 2 * 64 * 64 + 1 * 2 * 64 * 4


def test_argument_validation_messages(lib):
    from lowbit_quant_fa2_paddle_amd import _lib
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf)
    p = (p + 15) & ~15
    s = _lib.strides3((1024, 512, 64))
    # unsupported head_dim -> the reference's ValueError text (src/core.py:287)
    st = lib.lbfa_attn_fwd(p, p, p, 0, p, 0, None, p, p, None, 1, 2, 2, 8, 8, 96, s, s, s, s, 0, None)
    assert st == _lib.LBFA_EINVAL and b"Unsupported head_dim: 96" in lib.lbfa_last_error()
    with pytest.raises(ValueError, match="Unsupported head_dim: 96"):
        _lib.check(st, lib)
    # GQA divisibility (src/core.py docstring :257), causal needs Sq == Sk (causal forward :389)
    st = lib.lbfa_attn_fwd(p, p, p, 0, p, 0, None, p, p, None, 1, 3, 2, 8, 8, 64, s, s, s, s, 0, None)
    assert st == _lib.LBFA_EINVAL and b"divisible" in lib.lbfa_last_error()
    st = lib.lbfa_attn_fwd(p, p, p, 0, p, 0, None, p, p, None, 1, 2, 2, 8, 16, 64, s, s, s, s, 1, None)
    assert st == _lib.LBFA_EINVAL and b"qo_len and kv_len must be equal" in lib.lbfa_last_error()
    # fp8 V needs v_scale
    st = lib.lbfa_attn_fwd(p, p, p, 2, p, 0, None, p, p, None, 1, 2, 2, 8, 8, 64, s, s, None, s, 0, None)
    assert st == _lib.LBFA_EINVAL and b"v_scale" in lib.lbfa_last_error()
    # quantiser: dtype / qmax / blk
    st = lib.lbfa_quant_per_block(p, 5, None, 1, p, p, 1.0, 127, 128, 1, 1, 8, 64, s, s, None, 1, None, None)
    assert st == _lib.LBFA_EINVAL and b"float16 or bfloat16" in lib.lbfa_last_error()
    st = lib.lbfa_quant_per_block(p, 0, None, 1, p, p, 1.0, 15, 128, 1, 1, 8, 64, s, s, None, 1, None, None)
    assert st == _lib.LBFA_EINVAL and b"qmax" in lib.lbfa_last_error()
    st = lib.lbfa_quant_per_block(p, 0, None, 1, p, p, 1.0, 127, 32, 1, 1, 8, 64, s, s, None, 1, None, None)
    assert st == _lib.LBFA_EINVAL and b"blk" in lib.lbfa_last_error()
    st = lib.lbfa_mean_seq(p, 0, p, p, 0, 1, 1, 8, 64, s, None)
    assert st == _lib.LBFA_EINVAL and b"workspace" in lib.lbfa_last_error()


def test_one_call_and_varlen_validation(lib):
    """The one-call, packed-varlen and un-quantised entry points reject bad arguments before any launch."""
    from lowbit_quant_fa2_paddle_amd import _lib
    buf = ctypes.create_string_buffer(1 << 16)
    p = (ctypes.addressof(buf) + 255) & ~255
    s3, s2 = _lib.strides3((1024, 512, 64)), _lib.strides2((64, 128))
    # workspace sizes: pure functions; head dims that are multiples of 8 are accepted, others are not
    need = lib.lbfa_forward_workspace_bytes(1, 2, 2, 8, 8, 64, 0, 1, 0)
    assert need > 0 and lib.lbfa_forward_workspace_bytes(1, 2, 2, 8, 8, 80, 0, 1, 0) > need
    assert lib.lbfa_forward_workspace_bytes(1, 2, 2, 8, 8, 36, 0, 1, 0) == 0
    assert lib.lbfa_forward_varlen_workspace_bytes(2, 2, 2, 16, 16, 8, 8, 64) > 0
    # lbfa_forward
    st = lib.lbfa_forward(p, p, p, 0, p, None, p, need, 1, 2, 2, 8, 8, 36, s3, s3, s3, s3, 0.125, 127, 127, 0, 0, 1, None)
    assert st == _lib.LBFA_EINVAL and b"Unsupported head_dim: 36" in lib.lbfa_last_error()
    st = lib.lbfa_forward(p, p, p, 0, p, None, p, 16, 1, 2, 2, 8, 8, 64, s3, s3, s3, s3, 0.125, 127, 127, 0, 0, 1, None)
    assert st == _lib.LBFA_EINVAL and b"workspace too small" in lib.lbfa_last_error()
    st = lib.lbfa_forward(p, p, p, 0, p, None, p, need, 1, 3, 2, 8, 8, 64, s3, s3, s3, s3, 0.125, 127, 127, 0, 0, 1, None)
    assert st == _lib.LBFA_EINVAL and b"divisible" in lib.lbfa_last_error()
    # packed batches
    st = lib.lbfa_forward_varlen(p, p, p, 0, p, None, p, p, 1 << 15, 2, 2, 2, 16, 16, 8, 8, 64, s2, s2, s2, s2, 0.125, 127, 127, 0, 1, None)
    assert st == _lib.LBFA_EINVAL and b"null pointer" in lib.lbfa_last_error()
    st = lib.lbfa_forward_varlen(p, p, p, 0, p, p, p, p, 1 << 15, 2, 2, 2, 16, 16, 8, 8, 136, s2, s2, s2, s2, 0.125, 127, 127, 0, 1, None)
    assert st == _lib.LBFA_EINVAL and b"Unsupported head_dim: 136" in lib.lbfa_last_error()
    st = lib.lbfa_quant_per_block_varlen(p, 0, None, 1, p, p, p, None, 1.0, 127, 128, 2, 8, 2, 64, s2, s2, None)
    assert st == _lib.LBFA_EINVAL and b"null pointer" in lib.lbfa_last_error()
    st = lib.lbfa_attn_fwd_varlen(p, p, p, 2, p, 0, p, p, p, p, p, p, 2, 2, 2, 8, 8, 64, s2, s2, s2, s2, 0, None)
    assert st == _lib.LBFA_EINVAL and b"v must be float16" in lib.lbfa_last_error()
    # the attention entry points take fp16 (or e4m3) V, as the reference's kernel does: bf16 is cast first (src/core.py:307-308)
    st = lib.lbfa_attn_fwd_varlen(p, p, p, 1, p, 0, p, p, p, p, p, p, 2, 2, 2, 8, 8, 64, s2, s2, s2, s2, 0, None)
    assert st == _lib.LBFA_EINVAL and b"lbfa_cast_bf16_to_f16" in lib.lbfa_last_error()
    st = lib.lbfa_attn_fwd(p, p, p, 1, p, 0, None, p, p, None, 1, 2, 2, 8, 8, 64, s3, s3, s3, s3, 0, None)
    assert st == _lib.LBFA_EINVAL and b"lbfa_cast_bf16_to_f16" in lib.lbfa_last_error()
    st = lib.lbfa_cast_bf16_to_f16(p, p, 1, 2, 8, 36, s3, s3, None)
    assert st == _lib.LBFA_EINVAL and b"multiple of 8" in lib.lbfa_last_error()
    # the router's statistic (lbfa_absmax): null output, dtype, head dim, alignment
    st = lib.lbfa_absmax(p, 0, None, 1, 2, 8, 64, s3, None)
    assert st == _lib.LBFA_EINVAL and b"null pointer" in lib.lbfa_last_error()
    st = lib.lbfa_absmax(p, 2, p, 1, 2, 8, 64, s3, None)
    assert st == _lib.LBFA_EINVAL and b"float16 or bfloat16" in lib.lbfa_last_error()
    st = lib.lbfa_absmax(p, 0, p, 1, 2, 8, 36, s3, None)
    assert st == _lib.LBFA_EINVAL and b"multiple of 8" in lib.lbfa_last_error()
    st = lib.lbfa_absmax(p + 2, 0, p, 1, 2, 8, 64, s3, None)
    assert st == _lib.LBFA_EINVAL and b"16-byte aligned" in lib.lbfa_last_error()
    # a bf16 V is cast into the workspace (not for fp8 PV, whose V quantiser reads bf16 itself): the dtype-aware size is larger
    f16, bf16 = _lib.LBFA_F16, _lib.LBFA_BF16
    assert lib.lbfa_forward_workspace_bytes_dt(1, 2, 2, 8, 8, 128, bf16, 0, 1, 0) > lib.lbfa_forward_workspace_bytes_dt(1, 2, 2, 8, 8, 128, f16, 0, 1, 0)
    assert lib.lbfa_forward_workspace_bytes_dt(1, 2, 2, 8, 8, 64, bf16, 1, 1, 0) == lib.lbfa_forward_workspace_bytes_dt(1, 2, 2, 8, 8, 64, f16, 1, 1, 0)
    assert lib.lbfa_forward_workspace_bytes(1, 2, 2, 8, 8, 128, 0, 1, 0) == lib.lbfa_forward_workspace_bytes_dt(1, 2, 2, 8, 8, 128, bf16, 0, 1, 0)
    assert lib.lbfa_forward_varlen_workspace_bytes_dt(2, 2, 2, 16, 16, 8, 8, 128, bf16) > lib.lbfa_forward_varlen_workspace_bytes_dt(2, 2, 2, 16, 16, 8, 8, 128, f16)
    # un-quantised kernel
    st = lib.lbfa_sdpa_fwd(p, p, p, 0, p, None, 1, 2, 2, 8, 16, 64, s3, s3, s3, s3, 0.125, 1, None)
    assert st == _lib.LBFA_EINVAL and b"qo_len and kv_len must be equal" in lib.lbfa_last_error()
    st = lib.lbfa_sdpa_fwd(p, p, p, 2, p, None, 1, 2, 2, 8, 8, 64, s3, s3, s3, s3, 0.125, 0, None)
    assert st == _lib.LBFA_EINVAL and b"float16 or bfloat16" in lib.lbfa_last_error()
    st = lib.lbfa_sdpa_fwd(p, p, p, 0, p, None, 1, 2, 2, 8, 8, 64, s3, s3, s3, s3, -0.125, 0, None)
    assert st == _lib.LBFA_EINVAL and b"sm_scale must be positive" in lib.lbfa_last_error()


def test_api_surface_matches_reference():
    """Every name the reference package exports (src/__init__.py:1-17) is importable from the package root."""
    import lowbit_quant_fa2_paddle_amd as lb
    for name in ["sageattn", "sageattn_varlen", "sageattn_qk_int8_pv_fp16_triton", "sageattn_qk_int8_pv_fp16_cuda",
                 "sageattn_qk_int8_pv_fp8_cuda", "sageattn_qk_int4_pv_fp16_triton", "lowbit_fa_attn", "lowbit_fa_varlen",
                 "lowbit_fa_multi_precision", "lowbit_fa_qk_int8_pv_fp16_triton", "lowbit_fa_qk_int8_pv_fp16_cuda",
                 "lowbit_fa_qk_int8_pv_fp8_cuda", "lowbit_fa_qk_int4_pv_fp16_triton"]:
        assert callable(getattr(lb, name)), name
    assert lb.lowbit_fa_qk_int8_pv_fp16_triton is lb.sageattn_qk_int8_pv_fp16_triton


def test_host_checks_without_gpu():
    """Host-side checks of src/core.py:269-290 fire before any device work; CPU tensors are refused loudly."""
    import torch
    import lowbit_quant_fa2_paddle_amd as lb
    q = torch.zeros(1, 2, 16, 64, dtype=torch.float32)
    with pytest.raises(AssertionError, match="float16 or torch.bfloat16"):
        lb.lowbit_fa_qk_int8_pv_fp16_triton(q, q, q)
    q16 = q.half()
    with pytest.raises(AssertionError, match="same dtype"):
        lb.lowbit_fa_qk_int8_pv_fp16_triton(q16, q16, q16.bfloat16())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lb.lowbit_fa_qk_int8_pv_fp16_triton(q16, q16, q16)
    with pytest.raises(ValueError, match="Unsupported quantization backend"):
        lb.lowbit_fa_qk_int8_pv_fp16_triton(q16, q16, q16, quantization_backend="x")
    with pytest.raises(AssertionError, match="qk_quant_gran"):
        lb.lowbit_fa_qk_int8_pv_fp8_cuda(q16, q16, q16, qk_quant_gran="per_block")


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package may import it."""
    pkg = os.path.join(ROOT, "lowbit_quant_fa2_paddle_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dirpath, f)
