"""GPU tests of the un-quantised attention kernel (`lbfa_sdpa_fwd`, the FP16 branch of the precision router,
src/core.py:1066-1096) against fp32 SDPA on the same fp16 / bf16 inputs (oracle.sdpa_naive, src/core.py:46-69).

Tolerance: the kernel rounds P to the input dtype before the PV product (as FlashAttention-2 does: fp16 P x fp16 V, bf16 P x
bf16 V on the bf16 MFMA) and the output to fp16 / bf16: |dO| <= 2e-3 + 2e-3 |O| (bf16: 4e-3 + (2e-3 + 2^-7) |O|, P carries 2^-9); the row sums
ride on the same MFMA, i.e. over the rounded P: LSE <= 1e-3 (fp16; one fp16 rounding of a lone P is 4.9e-4) / 2.5e-3 (bf16: 2e-3)."""
import numpy as np
import pytest

from test_gpu_parity import TDT, _canon, _np, _o_close, _t, dev  # noqa: F401

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,layout,causal,B,H,Hkv,S,Sk,D", [
    ("fp16", "HND", False, 2, 4, 4, 256, 256, 64),
    ("fp16", "NHD", True, 1, 4, 2, 300, 300, 128),
    ("fp16", "HND", False, 1, 2, 1, 130, 333, 64),     # cross lengths, ragged key tail
    ("bf16", "NHD", True, 2, 2, 2, 200, 200, 64),
    ("bf16", "HND", False, 1, 2, 2, 64, 1000, 128),
    ("fp16", "NHD", False, 1, 2, 2, 77, 77, 80),       # head dim padded inside the kernel
    ("fp16", "HND", True, 1, 2, 2, 1, 1, 64),
])
def test_fp16_kernel_vs_fp32_sdpa(oracle, dev, dtype, layout, causal, B, H, Hkv, S, Sk, D):
    from lowbit_quant_fa2_paddle_amd import core
    q, k, v = oracle.make_inputs(B, H, S, D, seed=5, layout=layout, dtype=dtype, Hkv=Hkv, Sk=Sk, k_bias=0.5)
    tq, tk, tv = (_t(a, dtype, dev) for a in (q, k, v))
    o, lse = core.flash_attn_fp16(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True)
    assert o.dtype == TDT[dtype] and tuple(o.shape) == q.shape
    ref, rlse = oracle.sdpa_naive(*(_canon(a, layout).astype(np.float64) for a in (q, k, v)), is_causal=causal,
                                  sm_scale=D ** -0.5, return_lse=True)
    _o_close(_canon(_np(o), layout), ref, dtype, atol=2e-3 if dtype == "fp16" else 4e-3)
    assert np.abs(_np(lse) - rlse).max() <= (1e-3 if dtype == "fp16" else 2.5e-3)


def test_fp16_kernel_large_scores_and_router(oracle, dev):
    """Inputs scaled x8 (scores ~ +-500 in natural units): the online softmax must stay exact; the precision router
    sends such inputs to this kernel (avg max|x|/127 > 0.2, src/core.py:1051-1063)."""
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(1, 2, 512, 64, seed=1)
    q, k, v = (oracle.to_storage(a * 8, "fp16") for a in (q, k, v))
    t = [_t(a, "fp16", dev) for a in (q, k, v)]
    assert lb.core.select_quantization(*t) == "FP16"
    o = lb.lowbit_fa_multi_precision(*t, is_causal=True)
    ref = oracle.sdpa_naive(*(a.astype(np.float64) for a in (q, k, v)), is_causal=True)
    _o_close(_np(o), ref, "fp16", atol=2e-3 * 8, rtol=2e-3)
    assert torch.equal(o, lb.core.flash_attn_fp16(*t, is_causal=True))


def test_fp16_kernel_validation(dev):
    from lowbit_quant_fa2_paddle_amd import core
    x = torch.randn(1, 2, 64, 64, dtype=torch.float16, device=dev)
    with pytest.raises(ValueError, match="Unsupported head_dim"):
        core.flash_attn_fp16(torch.randn(1, 2, 64, 136, dtype=torch.float16, device=dev), x, x)
    with pytest.raises(AssertionError):
        core.flash_attn_fp16(x, x[:, :, :32], x[:, :, :32], is_causal=True)
    assert tuple(core.flash_attn_fp16(x[:, :, :0], x, x).shape) == (1, 2, 0, 64)


def test_bf16_inputs_keep_their_range(oracle, dev):
    """bf16 Q / K go to the bf16 MFMA unconverted: magnitudes beyond the fp16 range (here a 7e4 offset on a key
    channel the queries ignore) must not turn into inf * 0 = NaN."""
    from lowbit_quant_fa2_paddle_amd import core
    q, k, v = oracle.make_inputs(1, 2, 200, 64, seed=6, dtype="bf16")
    q[..., 0] = 0.0
    k[..., 0] = 70144.0  # exactly representable in bf16, > 65504
    tq, tk, tv = (_t(a, "bf16", dev) for a in (q, k, v))
    o = core.flash_attn_fp16(tq, tk, tv)
    assert torch.isfinite(o).all()
    ref = oracle.sdpa_naive(*(a.astype(np.float64) for a in (q, k, v)), sm_scale=64 ** -0.5)
    _o_close(_np(o), ref, "bf16", atol=4e-3)
