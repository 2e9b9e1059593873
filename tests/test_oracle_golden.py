"""Pins the CPU oracle (oracle/lowbit_fa_oracle.py) against golden vectors produced by the
reference's own Triton kernels under TRITON_INTERPRET=1 (tests/golden/make_golden.py).

Contract: quantiser codes and scales bit-exact; O within 1e-3 abs; LSE within 1e-5.
"""
import hashlib

import numpy as np
import pytest

from conftest import golden_inputs as _inputs, golden_names, load_golden, varlen_golden_names


@pytest.mark.parametrize("name", golden_names())
def test_inputs_reproducible(oracle, name):
    p, g = load_golden(name)
    q, k, v = _inputs(oracle, p)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (q, k, v))).hexdigest()
    assert digest == g["input_sha256"], "seeded input generator drifted from the one that made the fixtures"


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_kernels(oracle, name):
    p, g = load_golden(name)
    q, k, v = _inputs(oracle, p)
    o, lse, mid = oracle.lowbit_fa_forward(q, k, v, dtype=p["dtype"], tensor_layout=p["layout"], is_causal=p["causal"],
                                           smooth_k=p["smooth_k"], return_lse=True, q_qmax=p["q_qmax"],
                                           k_qmax=p["k_qmax"], return_intermediates=True)
    canon = (lambda a: a) if p["layout"] == "HND" else (lambda a: np.transpose(a, (0, 2, 1, 3)))
    if p["smooth_k"]:
        km_g = canon(g["km"])
        # mean over the sequence: summation order may differ by one storage-dtype ulp
        ulp = 2.0 ** -10 if p["dtype"] == "fp16" else 2.0 ** -7
        assert np.max(np.abs(mid["km"] - km_g)) <= ulp * max(1.0, np.max(np.abs(km_g)))
    # quantiser: bit-exact codes and scales (when km agrees exactly, which it does on these fixtures)
    if not p["smooth_k"] or np.array_equal(mid["km"], canon(g["km"])):
        assert np.array_equal(mid["q_i8"], canon(g["q_i8"]))
        assert np.array_equal(mid["k_i8"], canon(g["k_i8"]))
        assert np.array_equal(mid["q_scale"].view(np.uint32), g["q_scale"].view(np.uint32))
        assert np.array_equal(mid["k_scale"].view(np.uint32), g["k_scale"].view(np.uint32))
    else:
        pytest.fail("km differs from the fixture; quantiser pin not exercised")
    # attention: restatement vs interpreted reference kernel
    # fp16 out: 1e-3 abs.  bf16 out: additionally one bf16 ulp (2^-7 relative) for results that sit on
    # a rounding boundary of the coarser output grid.
    rtol = 2.0 ** -7 if p["dtype"] == "bf16" else 0.0
    assert np.all(np.abs(o - g["o"]) <= 1e-3 + rtol * np.abs(g["o"]))
    lse2 = g["lse2"]  # raw base-2 LSE of the kernel; the API converts it (src/core.py:344-350)
    D = p["D"]
    sm_scale = 1.0 / D ** 0.5
    lse_expect = lse2 / np.float32(1.44269504)
    if p["smooth_k"]:
        qc = canon(q)
        kmq = np.repeat(mid["km"], p["H"] // p["Hkv"], axis=1)
        Dp = kmq.shape[-1]
        qpad = np.pad(qc, [(0, 0)] * 3 + [(0, Dp - D)])
        corr = oracle.to_storage(np.einsum("bhsd,bhtd->bhs", qpad, kmq), p["dtype"])
        lse_expect = lse_expect + corr * np.float32(sm_scale)
    assert np.max(np.abs(lse - lse_expect)) <= 1e-5 * max(1.0, np.max(np.abs(lse_expect)))


@pytest.mark.parametrize("name", ["c1_hnd_s256_d64", "nhd_s512_d128_kbias_causal", "gqa_h4_kv2_s256_d64"])
def test_reference_accuracy_vs_fp32_sdpa(oracle, name):
    """Sanity: reference-kernel output vs fp32 SDPA, MSE <= 1e-5 on N(0,1) inputs."""
    p, g = load_golden(name)
    q, k, v = _inputs(oracle, p)
    canon = (lambda a: a) if p["layout"] == "HND" else (lambda a: np.transpose(a, (0, 2, 1, 3)))
    ref = oracle.sdpa_naive(canon(q), canon(k), canon(v), is_causal=p["causal"])
    mse = float(np.mean((canon(g["o"]) - ref) ** 2))
    assert mse <= 1e-5, mse


@pytest.mark.parametrize("name", varlen_golden_names())
def test_varlen_oracle_matches_reference_kernels(oracle, name):
    """Packed batches: the oracle's `lowbit_fa_varlen` vs the reference's varlen quantiser + attention kernels."""
    p, g = load_golden(name)
    q, k, v, cu_q, cu_k = oracle.make_varlen_inputs(p["lens_q"], p["lens_k"], p["Hq"], p["Hkv"], p["D"], seed=p["seed"],
                                                    dtype=p["dtype"], k_bias=p["k_bias"])
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (q, k, v))).hexdigest()
    assert digest == g["input_sha256"]
    o, mid = oracle.lowbit_fa_varlen(q, k, v, cu_q, cu_k, dtype=p["dtype"], is_causal=p["causal"], smooth_k=p["smooth_k"],
                                     return_intermediates=True)
    assert np.array_equal(mid["km"], g["km"]), "global mean differs from the fixture; quantiser pin not exercised"
    assert np.array_equal(mid["q_i8"], g["q_i8"])
    assert np.array_equal(mid["k_i8"], g["k_i8"])
    assert np.array_equal(mid["q_scale"].view(np.uint32), g["q_scale"].view(np.uint32))   # [sum_blocks, H]
    assert np.array_equal(mid["k_scale"].view(np.uint32), g["k_scale"].view(np.uint32))
    rtol = 2.0 ** -7 if p["dtype"] == "bf16" else 0.0
    assert np.all(np.abs(o - g["o"]) <= 1e-3 + rtol * np.abs(g["o"]))
