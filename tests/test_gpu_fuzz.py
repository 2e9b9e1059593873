"""Seeded fuzz of the public operators against the CPU oracle: random small shapes (ragged lengths, GQA, head dims,
layouts, dtypes, causal, return_lse, int8 / int4-range / fp8-PV / un-quantised, packed batches).  Catches shape-dependent
code paths (odd tile counts, single-tile sequences, masked instances ...) that hand-picked cases miss."""
import os

import numpy as np
import pytest

from test_gpu_parity import TDT, _canon, _np, _o_close, _t, dev  # noqa: F401

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


N_LOW_BIT = int(os.environ.get("LBFA_FUZZ_N", "40"))  # more seeds for a one-off hunt: LBFA_FUZZ_N=400 pytest tests/test_gpu_fuzz.py -m gpu
N_OTHER = max(12, N_LOW_BIT // 3)
SEED0 = int(os.environ.get("LBFA_FUZZ_SEED0", "0"))  # shifts the whole seed range for another hunt


def _rand_cfg(rng):
    D = int(rng.choice([64, 128, 32, 80, 96, 40, 36, 120]))
    Hkv = int(rng.choice([1, 2, 3]))
    H = Hkv * int(rng.choice([1, 2, 4]))
    causal = bool(rng.integers(0, 2))
    Sq = int(rng.choice([1, 17, 64, 65, 127, 128, 129, 191, 192, 200, 256, 300, 320, 383, 448, 513]))
    Sk = Sq if causal or rng.integers(0, 2) else int(rng.choice([1, 33, 64, 100, 192, 257, 320, 450]))
    return dict(B=int(rng.integers(1, 3)), H=H, Hkv=Hkv, Sq=Sq, Sk=Sk, D=D, causal=causal,
                layout=str(rng.choice(["HND", "NHD"])), dtype=str(rng.choice(["fp16", "bf16"])),
                lse=bool(rng.integers(0, 2)), smooth=bool(rng.integers(0, 4)), bias=float(rng.choice([0.0, 0.3, -0.5])),
                sm_scale=(None if rng.integers(0, 3) else float(rng.choice([0.07, 0.2, 1.0 / 3.0]))))


@pytest.mark.parametrize("seed", range(N_LOW_BIT))
def test_fuzz_low_bit_operators(oracle, dev, seed):
    import lowbit_quant_fa2_paddle_amd as lb
    rng = np.random.default_rng(1000 + SEED0 + seed)
    c = _rand_cfg(rng)
    kind = ["int8", "int8", "int4", "q8k4", "fp8"][seed % 5]
    q, k, v = oracle.make_inputs(c["B"], c["H"], c["Sq"], c["D"], seed=seed, layout=c["layout"], dtype=c["dtype"], Hkv=c["Hkv"],
                                 Sk=c["Sk"], k_bias=c["bias"])
    tq, tk, tv = (_t(a, c["dtype"], dev) for a in (q, k, v))
    kw = dict(tensor_layout=c["layout"], is_causal=c["causal"], smooth_k=c["smooth"], return_lse=c["lse"], sm_scale=c["sm_scale"])
    okw = dict(dtype=c["dtype"], tensor_layout=c["layout"], is_causal=c["causal"], smooth_k=c["smooth"], return_lse=c["lse"],
               sm_scale=c["sm_scale"], tail="neg_inf", amax_floor=1e-7)
    if kind == "int8":
        out, ref = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, **kw), oracle.lowbit_fa_forward(q, k, v, **okw)
    elif kind == "int4":
        out, ref = lb.lowbit_fa_qk_int4_pv_fp16_triton(tq, tk, tv, **kw), oracle.lowbit_fa_forward(q, k, v, q_qmax=7, k_qmax=7, **okw)
    elif kind == "q8k4":
        out, ref = lb.lowbit_fa_qk_int4_pv_fp16_triton(tq, tk, tv, q_bits=8, **kw), oracle.lowbit_fa_forward(q, k, v, k_qmax=7, **okw)
    else:
        okw.pop("tail")
        out, ref = lb.lowbit_fa_qk_int8_pv_fp8_cuda(tq, tk, tv, **kw), oracle.lowbit_fa_forward(q, k, v, pv="fp8", **okw)
    o, o_ref = (out[0], ref[0]) if c["lse"] else (out, ref)
    assert tuple(o.shape) == q.shape, c
    if kind == "fp8":
        # e4m3 P has 3 mantissa bits: a 1-ulp exp2 difference at a rounding boundary flips a code (6 % of that P), which
        # shows in rows with few keys (parity unpinned for this path) - loose element bound, tight mean-square bound
        # (a causal row with k keys moves by up to ~6 % * |v| / sqrt(k) per flipped code: up to 0.1 for the first rows)
        _o_close(_np(o), o_ref, c["dtype"], atol=0.2, rtol=6e-2)
        assert float(np.mean((_np(o) - o_ref) ** 2)) <= 2e-5, c
    else:
        _o_close(_np(o), o_ref, c["dtype"])
    if c["lse"]:
        # the smooth-K correction q.km is rounded to the storage dtype (src/core.py:294-304): one ulp of it, times sm_scale,
        # is the granularity of the LSE (summation order differs between any two implementations)
        tol = 2e-3 + 2.0 ** -9 * float(np.abs(ref[1]).max())
        assert np.abs(_np(out[1]) - ref[1]).max() <= tol, c


@pytest.mark.parametrize("seed", range(N_OTHER))
def test_fuzz_unquantised_and_varlen(oracle, dev, seed):
    import lowbit_quant_fa2_paddle_amd as lb
    rng = np.random.default_rng(2000 + SEED0 + seed)
    c = _rand_cfg(rng)
    if seed % 2 == 0:  # un-quantised kernel vs fp64 SDPA
        q, k, v = oracle.make_inputs(c["B"], c["H"], c["Sq"], c["D"], seed=seed, layout=c["layout"], dtype=c["dtype"], Hkv=c["Hkv"],
                                     Sk=c["Sk"], k_bias=c["bias"])
        tq, tk, tv = (_t(a, c["dtype"], dev) for a in (q, k, v))
        o = lb.core.flash_attn_fp16(tq, tk, tv, tensor_layout=c["layout"], is_causal=c["causal"], sm_scale=c["sm_scale"])
        ref = oracle.sdpa_naive(*(_canon(a, c["layout"]).astype(np.float64) for a in (q, k, v)), is_causal=c["causal"],
                                sm_scale=c["sm_scale"] or c["D"] ** -0.5)
        # bf16 inputs: P and V stay bf16 on the bf16 MFMA (as a bf16 FlashAttention-2): P carries 2^-9, so an output can land one
        # bf16 ulp off even at the bottom of its binade (2^-7 of the value) - twice the absolute slack of the fp16 kernel
        _o_close(_canon(_np(o), c["layout"]), ref, c["dtype"], atol=4e-3 if c["dtype"] == "bf16" else 2e-3)
    else:  # packed batch vs the varlen oracle
        n = int(rng.integers(1, 5))
        lens_q = [int(x) for x in rng.choice([1, 5, 64, 100, 128, 129, 250, 320], size=n)]
        lens_k = lens_q if c["causal"] else [int(x) for x in rng.choice([1, 40, 64, 65, 192, 300], size=n)]
        q, k, v, cu_q, cu_k = oracle.make_varlen_inputs(lens_q, lens_k, c["H"], c["Hkv"], c["D"], seed=seed, dtype=c["dtype"], k_bias=c["bias"])
        tq, tk, tv = (_t(a, c["dtype"], dev) for a in (q, k, v))
        o = lb.sageattn_varlen(tq, tk, tv, torch.from_numpy(cu_q).to(dev), torch.from_numpy(cu_k).to(dev), max(lens_q), max(lens_k),
                               is_causal=c["causal"], smooth_k=c["smooth"], sm_scale=c["sm_scale"])
        ref = oracle.lowbit_fa_varlen(q, k, v, cu_q, cu_k, dtype=c["dtype"], is_causal=c["causal"], smooth_k=c["smooth"], tail="neg_inf",
                                      amax_floor=1e-7, sm_scale=c["sm_scale"])
        _o_close(_np(o), ref, c["dtype"])


@pytest.mark.parametrize("seed", range(max(10, N_LOW_BIT // 4)))
def test_fuzz_modular_entry_points(oracle, dev, seed):
    """mean_seq -> quantize (Q, K) -> attention through the modular C-ABI entry points on random shapes and on strided
    (packed-qkv) views: km, codes and scales bit-exact against the oracle's intermediates, O within tolerance."""
    from lowbit_quant_fa2_paddle_amd import attn_qk_int8_per_block as attn, quant_per_block as qpb
    rng = np.random.default_rng(3000 + SEED0 + seed)
    c = _rand_cfg(rng)
    D = int(rng.choice([64, 128]))  # the modular entry points take padded head dims only
    qm, km_ = [(127, 127), (7, 7), (127, 7)][seed % 3]
    q, k, v = oracle.make_inputs(c["B"], c["H"], c["Sq"], D, seed=seed, layout=c["layout"], dtype=c["dtype"], Hkv=c["Hkv"], Sk=c["Sk"],
                                 k_bias=c["bias"])
    tq, tk, tv = (_t(a, c["dtype"], dev) for a in (q, k, v))
    if seed % 2 and c["Sq"] == c["Sk"] and c["H"] == c["Hkv"] and c["layout"] == "NHD":  # strided views of one packed tensor
        qkv = torch.stack([tq, tk, tv], dim=2)
        tq, tk, tv = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    sm = c["sm_scale"] or D ** -0.5
    o_ref, lse_ref, mid = oracle.lowbit_fa_forward(q, k, v, dtype=c["dtype"], tensor_layout=c["layout"], is_causal=c["causal"],
                                                   smooth_k=c["smooth"], sm_scale=sm, q_qmax=qm, k_qmax=km_, return_lse=True,
                                                   tail="neg_inf", amax_floor=1e-7, return_intermediates=True)
    km = qpb.mean_seq(tk, c["layout"]) if c["smooth"] else None
    if km is not None:
        assert np.array_equal(_np(km).reshape(mid["km"].shape), mid["km"])
    q8, qs = qpb.quantize(tq, sm_scale=sm * 1.44269504, qmax=qm, blk=128, tensor_layout=c["layout"])
    k8, ks = qpb.quantize(tk, sm_scale=1.0, qmax=km_, blk=64, tensor_layout=c["layout"], mean=km)
    assert np.array_equal(_canon(q8.cpu().numpy(), c["layout"]), mid["q_i8"]) and np.array_equal(_canon(k8.cpu().numpy(), c["layout"]), mid["k_i8"])
    assert np.array_equal(qs.cpu().numpy().view(np.uint32), mid["q_scale"].view(np.uint32))
    assert np.array_equal(ks.cpu().numpy().view(np.uint32), mid["k_scale"].view(np.uint32))
    o, _ = attn.forward(q8, k8, tv, qs, ks, tensor_layout=c["layout"], output_dtype=TDT[c["dtype"]], is_causal=c["causal"])
    _o_close(_np(o), o_ref, c["dtype"])
