"""GPU parity tests of the packed variable-length path (`sageattn_varlen`, src/core.py:356-491): the HIP kernels
behind lbfa_quant_per_block_varlen / lbfa_attn_fwd_varlen / lbfa_forward_varlen against the golden vectors of the
reference's varlen Triton kernels and against the CPU oracle.  Same bars as tests/test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import load_golden, varlen_golden_names
from test_gpu_parity import TDT, _np, _o_close, _t, dev  # noqa: F401  (dev is a fixture)

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _case(oracle, p, dev):
    q, k, v, cu_q, cu_k = oracle.make_varlen_inputs(p["lens_q"], p["lens_k"], p["Hq"], p["Hkv"], p["D"], seed=p["seed"],
                                                    dtype=p["dtype"], k_bias=p["k_bias"])
    t = [_t(a, p["dtype"], dev) for a in (q, k, v)]
    return (q, k, v, cu_q, cu_k), t, torch.from_numpy(cu_q).to(dev), torch.from_numpy(cu_k).to(dev)


@pytest.mark.parametrize("name", varlen_golden_names())
def test_varlen_quantiser_and_attention_vs_reference_golden(oracle, dev, name):
    """Modular entry points in the reference's layouts: codes / scales [sum_blocks, H] bit-exact against the
    reference varlen quantiser, then the attention kernel fed with the golden codes against the reference O."""
    from lowbit_quant_fa2_paddle_amd import attn_qk_int8_block_varlen as attn, quant_per_block_varlen as qv
    p, g = load_golden(name)
    (q, k, v, cu_q, cu_k), (tq, tk, tv), tcq, tck = _case(oracle, p, dev)
    km = torch.from_numpy(g["km"]).to(TDT[p["dtype"]]).to(dev)  # [1, Hkv, D], the reference's k.mean(dim=0)
    q8, qs, k8, ks, cqs, cks = qv.per_block_int8(tq, tk, tcq, tck, max(p["lens_q"]), max(p["lens_k"]),
                                                 sm_scale=p["D"] ** -0.5, km=km)
    assert np.array_equal(cqs.cpu().numpy(), g["cu_q_scale"]) and np.array_equal(cks.cpu().numpy(), g["cu_k_scale"])
    assert np.array_equal(q8.cpu().numpy(), g["q_i8"])
    assert np.array_equal(k8.cpu().numpy(), g["k_i8"])
    assert np.array_equal(qs.cpu().numpy().view(np.uint32), g["q_scale"].view(np.uint32))
    assert np.array_equal(ks.cpu().numpy().view(np.uint32), g["k_scale"].view(np.uint32))
    gq8, gk8 = torch.from_numpy(g["q_i8"]).to(dev), torch.from_numpy(g["k_i8"]).to(dev)
    gqs, gks = torch.from_numpy(g["q_scale"]).to(dev), torch.from_numpy(g["k_scale"]).to(dev)
    o = attn.forward(gq8, gk8, tv, tcq, tck, max(p["lens_q"]), gqs, gks, cqs, cks, output_dtype=TDT[p["dtype"]],
                     is_causal=p["causal"])
    _o_close(_np(o), g["o"], p["dtype"])


@pytest.mark.parametrize("name", varlen_golden_names())
def test_varlen_operator_vs_reference_golden(oracle, dev, name):
    import lowbit_quant_fa2_paddle_amd as lb
    p, g = load_golden(name)
    _, (tq, tk, tv), tcq, tck = _case(oracle, p, dev)
    o = lb.lowbit_fa_varlen(tq, tk, tv, tcq, tck, max(p["lens_q"]), max(p["lens_k"]), is_causal=p["causal"])
    assert o.dtype == TDT[p["dtype"]] and tuple(o.shape) == g["o"].shape
    _o_close(_np(o), g["o"], p["dtype"])


@pytest.mark.parametrize("dtype,D,Hq,Hkv,causal,lens_q,lens_k", [
    ("fp16", 64, 4, 2, False, [1, 300, 77, 128, 5], [33, 190, 1, 257, 64]),     # ragged key tails, 1-token sequences
    ("fp16", 128, 2, 2, True, [129, 64, 1, 200], [129, 64, 1, 200]),
    ("bf16", 64, 6, 2, True, [513, 31], [513, 31]),
    ("fp16", 80, 2, 1, False, [70, 140], [90, 60]),                               # head-dim pad (src/core.py:431-440)
])
def test_varlen_operator_vs_oracle(oracle, dev, dtype, D, Hq, Hkv, causal, lens_q, lens_k):
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v, cu_q, cu_k = oracle.make_varlen_inputs(lens_q, lens_k, Hq, Hkv, D, seed=21, dtype=dtype, k_bias=0.4)
    tq, tk, tv = (_t(a, dtype, dev) for a in (q, k, v))
    tcq, tck = torch.from_numpy(cu_q).to(dev), torch.from_numpy(cu_k.astype(np.int64)).to(dev)  # int32 and int64 tables
    o = lb.sageattn_varlen(tq, tk, tv, tcq, tck, max(lens_q), max(lens_k), is_causal=causal)
    ref = oracle.lowbit_fa_varlen(q, k, v, cu_q, cu_k, dtype=dtype, is_causal=causal, tail="neg_inf", amax_floor=1e-7)
    _o_close(_np(o), ref, dtype)
    # no sequence leaks into its neighbours: against fp32 SDPA per sequence
    for b in range(len(lens_q)):
        qb, kb, vb = (np.transpose(a[c[b]:c[b + 1]], (1, 0, 2))[None] for a, c in ((q, cu_q), (k, cu_k), (v, cu_k)))
        sd = oracle.sdpa_naive(qb, kb, vb, is_causal=causal, sm_scale=D ** -0.5)
        got = np.transpose(_np(o[cu_q[b]:cu_q[b + 1]]), (1, 0, 2))[None]
        assert float(np.mean((got - sd) ** 2)) <= 1e-4


def test_varlen_one_call_equals_modular_entry_points(oracle, dev):
    """lbfa_forward_varlen (padded internal scale layout) == mean + lbfa_quant_per_block_varlen x 2 +
    lbfa_attn_fwd_varlen (reference scale layout) composed on the host: bit-identical O."""
    import lowbit_quant_fa2_paddle_amd as lb
    from lowbit_quant_fa2_paddle_amd import attn_qk_int8_block_varlen as attn, quant_per_block as qpb, quant_per_block_varlen as qv
    lens_q, lens_k = [200, 64, 333], [100, 640, 333]
    q, k, v, cu_q, cu_k = oracle.make_varlen_inputs(lens_q, lens_k, 4, 2, 128, seed=3, k_bias=0.2)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    tcq, tck = torch.from_numpy(cu_q).to(dev), torch.from_numpy(cu_k).to(dev)
    o = lb.sageattn_varlen(tq, tk, tv, tcq, tck, max(lens_q), max(lens_k))
    km = qpb.mean_seq(tk[None], "NHD")[0]  # [1, Hkv, D]: mean over all packed tokens
    q8, qs, k8, ks, cqs, cks = qv.per_block_int8(tq, tk, tcq, tck, max(lens_q), max(lens_k), sm_scale=128 ** -0.5, km=km)
    o2 = attn.forward(q8, k8, tv, tcq, tck, max(lens_q), qs, ks, cqs, cks, max_seqlen_k=max(lens_k))
    assert torch.equal(o, o2)


def test_varlen_equals_dense_for_equal_lengths(oracle, dev):
    """A packed batch of ONE sequence is the dense NHD operator with B = 1 (same mean, same blocks): bit-identical."""
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(1, 4, 384, 64, seed=9, layout="NHD", k_bias=0.3)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    cu = torch.tensor([0, 384], dtype=torch.int32, device=dev)
    for causal in (False, True):
        o_d = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, tensor_layout="NHD", is_causal=causal)
        o_v = lb.sageattn_varlen(tq[0], tk[0], tv[0], cu, cu, 384, 384, is_causal=causal)
        assert torch.equal(o_d[0], o_v)


def test_varlen_validation(dev):
    import lowbit_quant_fa2_paddle_amd as lb
    cu = torch.tensor([0, 64], dtype=torch.int32, device=dev)
    x = torch.randn(64, 2, 64, dtype=torch.float16, device=dev)
    with pytest.raises(ValueError, match="Unsupported head_dim"):
        big = torch.randn(64, 2, 160, dtype=torch.float16, device=dev)
        lb.sageattn_varlen(big, big, big, cu, cu, 64, 64)
    with pytest.raises(ValueError, match="divisible"):
        lb.sageattn_varlen(torch.randn(64, 3, 64, dtype=torch.float16, device=dev), x, x, cu, cu, 64, 64)
    with pytest.raises(AssertionError):
        lb.sageattn_varlen(x.float(), x.float(), x.float(), cu, cu, 64, 64)
    assert tuple(lb.sageattn_varlen(x[:0], x, x, cu, cu, 64, 64).shape) == (0, 2, 64)


def test_varlen_empty_sequences(oracle, dev):
    """Zero-length sequences inside a packed batch: an empty query range produces nothing, an empty key range zeros
    (the reference starts from l = 1, acc = 0: attn_qk_int8_block_varlen.py:171-173,195)."""
    import lowbit_quant_fa2_paddle_amd as lb
    lens_q, lens_k = [64, 0, 100, 50], [64, 0, 100, 0]
    q, k, v, cu_q, cu_k = oracle.make_varlen_inputs(lens_q, lens_k, 2, 2, 64, seed=8, k_bias=0.2)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    o = lb.sageattn_varlen(tq, tk, tv, torch.from_numpy(cu_q).to(dev), torch.from_numpy(cu_k).to(dev), max(lens_q), max(lens_k))
    ref = oracle.lowbit_fa_varlen(q, k, v, cu_q, cu_k, tail="neg_inf", amax_floor=1e-7)
    _o_close(_np(o), ref, "fp16")
    assert float(np.abs(_np(o)[cu_q[3]:]).max()) == 0.0
