"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI via the Python
front end, against the CPU oracle and the committed golden vectors of the reference's Triton kernels.

Bars:
  * quantiser codes and scales: bit-exact;
  * km (mean over the sequence): bit-exact (fp64 accumulation on the device = the oracle's float64 mean);
  * attention O: |dO| <= 2e-3 + 2e-3*|O| vs the golden reference output / oracle (fp16 out; the kernel
    accumulates PV in fp32 where the reference rounds each 64-key tile product to fp16, and v_exp_f32
    is a ~1-ulp approximation); bf16 out adds one bf16 ulp (2^-7 relative);
  * LSE: <= 1e-3 abs;
  * sanity vs fp32 SDPA: MSE <= 1e-5 on N(0,1) inputs (int8), <= 1e-4 (fp8 PV, int4).
"""
import numpy as np
import pytest

from conftest import golden_inputs, golden_names, load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lowbit_quant_fa2_paddle_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _t(a, dtype, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(TDT[dtype]).to(dev)


def _np(t):
    return t.detach().float().cpu().numpy()


def _canon(a, layout):
    return a if layout == "HND" else np.transpose(a, (0, 2, 1, 3))


def _o_close(o, ref, dtype, atol=2e-3, rtol=2e-3):
    if dtype == "bf16":
        rtol = rtol + 2.0 ** -7
    err = np.abs(o - ref)
    bad = err > atol + rtol * np.abs(ref)
    assert not bad.any(), f"max err {err.max():.3e} at {np.argwhere(bad)[:3].tolist()} ({bad.sum()} elements)"


def _golden_o_close(o, ref, v, p):
    """Kernel output against a fixture made by the reference's kernels: |dO| <= 2e-3 + 2e-3 |O| on EVERY element, the N(0,1)
    fixtures and the randint / peaky ones alike (round 4: a Q block that leaves the lazy pass dequantises with the un-rounded
    q_scale * k_scale, attn_fwd16.hip `wide`, as attn_qk_int8_per_block.py:51 does)."""
    return _o_close(o, ref, p["dtype"])


def _fp8_close(o, ref):
    """fp8-PV against the oracle's restatement (parity unpinned): per element |dO| <= 1e-2 + 2e-2 |ref| (e4m3 P has 3
    mantissa bits; a 1-ulp difference of exp2 at a rounding boundary flips a code, 6 % of that P).  Round 3 had widened this to
    0.1 + 6e-2 |ref| with aggregate bounds for an experimental kernel that did not ship; every fp8 case of the suite meets the
    per-element bound on the shipped kernel (round 4: 424 GPU tests green with it)."""
    err = np.abs(o - ref)
    bad = err > 1e-2 + 2e-2 * np.abs(ref)
    assert not bad.any(), f"max err {err.max():.3e} at {np.argwhere(bad)[:3].tolist()} ({bad.sum()} elements)"


# ------------------------------------------------------------------------------------------------------
# quantiser
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("S,D", [(256, 64), (300, 128), (64, 64), (1, 64), (129, 128)])
def test_mean_and_quant_bit_exact(oracle, dev, layout, dtype, S, D):
    from lowbit_quant_fa2_paddle_amd import quant_per_block as qpb
    B, H = 2, 3
    q, k, v = oracle.make_inputs(B, H, S, D, seed=11, layout=layout, dtype=dtype, k_bias=0.3)
    tq, tk = _t(q, dtype, dev), _t(k, dtype, dev)
    km = qpb.mean_seq(tk, layout)
    km_np = _np(km).reshape(B, H, 1, D)
    km_ref = oracle.mean_seq(_canon(k, layout), dtype)
    # fp64 accumulation on the device: the mean is the correctly rounded one, bit-identical to the oracle's float64 mean
    assert np.array_equal(km_np, km_ref), np.abs(km_np - km_ref).max()
    sm_scale = 1.0 / D ** 0.5
    q8, qs, k8, ks = qpb.per_block_int8(tq, tk, km=km, sm_scale=sm_scale, tensor_layout=layout)
    # oracle fed with the km the device produced: everything downstream is bit-exact
    rq8, rqs, rk8, rks = oracle.per_block_int8(_canon(q, layout), _canon(k, layout), km_np, sm_scale, dtype, amax_floor=1e-7)
    assert np.array_equal(_canon(q8.cpu().numpy(), layout), rq8)
    assert np.array_equal(_canon(k8.cpu().numpy(), layout), rk8)
    assert np.array_equal(qs.cpu().numpy().view(np.uint32), rqs.view(np.uint32))
    assert np.array_equal(ks.cpu().numpy().view(np.uint32), rks.view(np.uint32))


@pytest.mark.parametrize("fn,qm,km_", [("per_block_int4_unpack", 7, 7), ("per_block_q_int8_k_int4", 127, 7)])
def test_int4_range_quant_bit_exact(oracle, dev, fn, qm, km_):
    from lowbit_quant_fa2_paddle_amd import quant_per_block as qpb
    q, k, v = oracle.make_inputs(1, 2, 256, 64, seed=5)
    tq, tk = _t(q, "fp16", dev), _t(k, "fp16", dev)
    q8, qs, k8, ks = getattr(qpb, fn)(tq, tk, sm_scale=0.125)
    rq8, rqs, rk8, rks = oracle.per_block_int8(q, k, None, 0.125, "fp16", q_qmax=qm, k_qmax=km_, amax_floor=1e-7)
    assert np.array_equal(q8.cpu().numpy(), rq8) and np.array_equal(k8.cpu().numpy(), rk8)
    assert np.array_equal(qs.cpu().numpy().view(np.uint32), rqs.view(np.uint32))
    assert np.array_equal(ks.cpu().numpy().view(np.uint32), rks.view(np.uint32))
    assert np.abs(k8.cpu().numpy()).max() <= 7


def test_quant_degenerate_blocks(oracle, dev):
    """All-zero block: the reference Triton kernel yields NaN (no epsilon); the HIP path floors amax at 1e-7
    like the CUDA quantiser (fused.cu:147) and emits zeros.  A block holding one huge value saturates at qmax."""
    from lowbit_quant_fa2_paddle_amd import quant_per_block as qpb
    x = torch.zeros(1, 1, 256, 64, dtype=torch.float16, device=dev)
    x[0, 0, 130, 5] = 60000.0
    x[0, 0, 131, 6] = -60000.0
    c, s = qpb.quantize(x, sm_scale=1.0, qmax=127, blk=128)
    c, s = c.cpu().numpy(), s.cpu().numpy()
    assert np.all(c[0, 0, :128] == 0) and s[0, 0, 0] == np.float32(1e-7) / np.float32(127)
    assert c[0, 0, 130, 5] == 127 and c[0, 0, 131, 6] == -127 and np.isfinite(s).all()


# ------------------------------------------------------------------------------------------------------
# attention kernel on the reference's own codes/scales (golden fixtures)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names())
def test_attention_kernel_vs_reference_golden(oracle, dev, name):
    """Feed the golden int8 codes + scales (made by the reference's quantiser kernels) to lbfa_attn_fwd and
    compare with the reference attention kernel's O and LSE."""
    from lowbit_quant_fa2_paddle_amd import attn_qk_int8_per_block as attn
    p, g = load_golden(name)
    q, k, v = golden_inputs(oracle, p)
    Dp = g["q_i8"].shape[-1]
    if Dp != p["D"]:
        v = np.pad(v, [(0, 0)] * 3 + [(0, Dp - p["D"])])
    q8 = torch.from_numpy(g["q_i8"]).to(dev)
    k8 = torch.from_numpy(g["k_i8"]).to(dev)
    qs = torch.from_numpy(g["q_scale"]).to(dev)
    ks = torch.from_numpy(g["k_scale"]).to(dev)
    tv = _t(v, p["dtype"], dev)
    o, lse = attn.forward(q8, k8, tv, qs, ks, tensor_layout=p["layout"], output_dtype=TDT[p["dtype"]],
                          return_lse=True, is_causal=p["causal"])
    _golden_o_close(_np(o)[..., :p["D"]], g["o"], v, p)
    # randint fixtures: |lse2| ~ 1e4, and the kernel's per-tile scale sits on a 2^-19-relative grid
    assert np.abs(lse.cpu().numpy() - g["lse2"]).max() <= 1e-3 + 2.0 ** -18 * np.abs(g["lse2"]).max()


# ------------------------------------------------------------------------------------------------------
# full operator vs oracle
# ------------------------------------------------------------------------------------------------------
CASES = [
    # B, H, Hkv, Sq, Sk, D, layout, causal, dtype
    (1, 2, 2, 256, 256, 64, "HND", False, "fp16"),       # BASELINE config 1
    (1, 2, 2, 256, 256, 64, "HND", True, "fp16"),
    (2, 4, 2, 384, 384, 128, "NHD", True, "fp16"),       # GQA + NHD + causal (config 3 shape family)
    (1, 2, 1, 200, 333, 64, "HND", False, "fp16"),       # ragged Sq / Sk
    (1, 2, 2, 333, 333, 128, "NHD", True, "fp16"),       # ragged causal
    (1, 3, 3, 1, 77, 64, "HND", False, "fp16"),          # single query row
    (1, 2, 2, 512, 512, 80, "HND", False, "fp16"),       # head-dim pad path 80 -> 128
    (1, 2, 2, 256, 256, 32, "NHD", True, "bf16"),        # head-dim pad 32 -> 64, bf16 in/out
    (1, 2, 2, 640, 640, 128, "HND", False, "bf16"),
]


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D,layout,causal,dtype", CASES)
def test_operator_int8_fp16_vs_oracle(oracle, dev, B, H, Hkv, Sq, Sk, D, layout, causal, dtype):
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(B, H, Sq, D, seed=21, layout=layout, dtype=dtype, Hkv=Hkv, Sk=Sk, k_bias=0.5)
    tq, tk, tv = (_t(a, dtype, dev) for a in (q, k, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True)
    assert o.shape == tq.shape and o.dtype == tq.dtype and tuple(lse.shape) == (B, H, Sq)
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, k, v, dtype=dtype, tensor_layout=layout, is_causal=causal,
                                              return_lse=True, tail="neg_inf", amax_floor=1e-7)
    _o_close(_np(o), o_ref, dtype)
    # lse_correction (q . km) is rounded to the storage dtype by the reference (src/core.py:294-304): allow one
    # ulp of that rounding on top of the 1e-3 kernel tolerance
    corr_ulp = (2.0 ** -10 if dtype == "fp16" else 2.0 ** -7) * np.abs(lse_ref).max()
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 1e-3 + corr_ulp
    # sanity vs exact attention
    ref = oracle.sdpa_naive(_canon(q, layout).astype(np.float64), _canon(k, layout).astype(np.float64),
                            _canon(v, layout).astype(np.float64), is_causal=causal)
    mse = float(np.mean((_canon(_np(o), layout) - ref) ** 2))
    assert mse <= (1e-5 if dtype == "fp16" else 3e-5), mse


@pytest.mark.parametrize("smooth_k", [True, False])
@pytest.mark.parametrize("return_lse", [True, False])
def test_operator_flags(oracle, dev, smooth_k, return_lse):
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(1, 2, 256, 64, seed=3, k_bias=1.0)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    out = lb.sageattn_qk_int8_pv_fp16_cuda(tq, tk, tv, smooth_k=smooth_k, return_lse=return_lse, sm_scale=0.1,
                                           pv_accum_dtype="fp16+fp32", qk_quant_gran="per_warp")
    ref = oracle.lowbit_fa_forward(q, k, v, smooth_k=smooth_k, return_lse=return_lse, sm_scale=0.1, amax_floor=1e-7)
    if return_lse:
        _o_close(_np(out[0]), ref[0], "fp16")
        assert np.abs(out[1].cpu().numpy() - ref[1]).max() <= 2e-3
    else:
        _o_close(_np(out), ref, "fp16")


@pytest.mark.parametrize("q_bits", [4, 8])
@pytest.mark.parametrize("causal", [False, True])
def test_operator_int4(oracle, dev, q_bits, causal):
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(1, 2, 512, 64, seed=9)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    o = lb.lowbit_fa_qk_int4_pv_fp16_triton(tq, tk, tv, is_causal=causal, q_bits=q_bits)
    o_ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, q_qmax=7 if q_bits == 4 else 127, k_qmax=7, amax_floor=1e-7)
    _o_close(_np(o), o_ref, "fp16")
    ref = oracle.sdpa_naive(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), is_causal=causal)
    assert float(np.mean((_np(o) - ref) ** 2)) <= 5e-3  # 4-bit codes: coarse by construction


@pytest.mark.parametrize("B,H,Hkv,S,D,layout,causal,dtype", [
    (1, 2, 2, 256, 64, "HND", False, "fp16"),
    (2, 4, 2, 384, 128, "NHD", True, "fp16"),
    (1, 2, 2, 333, 128, "HND", False, "bf16"),
    (1, 2, 2, 200, 64, "NHD", True, "fp16"),
    # odd numbers of 64-key tiles (the block-scaled MFMA's result was once read too early on that path)
    (1, 2, 2, 64, 64, "HND", False, "fp16"),
    (2, 2, 1, 192, 128, "HND", False, "fp16"),
    (1, 2, 2, 320, 128, "NHD", False, "bf16"),
])
def test_operator_int8_fp8_vs_oracle(oracle, dev, B, H, Hkv, S, D, layout, causal, dtype):
    """fp8-PV: the oracle restates CUDA code that cannot run here and the reference holds no fixture -
    parity unpinned; checked against that restatement (loosely: e4m3 P has 3 mantissa bits, and exp2
    rounding differences flip fp8 codes) and against fp32 SDPA."""
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(B, H, S, D, seed=31, layout=layout, dtype=dtype, Hkv=Hkv, k_bias=0.25)
    tq, tk, tv = (_t(a, dtype, dev) for a in (q, k, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp8_cuda(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True)
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, k, v, dtype=dtype, tensor_layout=layout, is_causal=causal,
                                              return_lse=True, pv="fp8", amax_floor=1e-7)
    _fp8_close(_np(o), o_ref)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 2e-3 + 2.0 ** -9 * np.abs(lse_ref).max()
    ref = oracle.sdpa_naive(_canon(q, layout).astype(np.float64), _canon(k, layout).astype(np.float64),
                            _canon(v, layout).astype(np.float64), is_causal=causal)
    mse = float(np.mean((_canon(_np(o), layout) - ref) ** 2))
    assert mse <= 1e-4, mse


def test_v_fp8_quant_exact(oracle, dev):
    """per_channel_fp8: scales bit-exact, e4m3 codes equal to the numpy e4m3fn RNE-saturating encoder."""
    from lowbit_quant_fa2_paddle_amd import quant
    B, H, S, D = 1, 2, 200, 64
    _, _, v = oracle.make_inputs(B, H, S, D, seed=13)
    tv = _t(v, "fp16", dev)
    v8, vs, vm = quant.per_channel_fp8(tv)
    assert vm is None
    ref8, ref_s = oracle.per_channel_fp8(v)
    assert np.array_equal(vs.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))
    from fp8_layout import decode_v_fp8
    codes = oracle.e4m3fn_encode(ref8)  # [B,H,S,D]
    got = decode_v_fp8(v8.buf.cpu().numpy(), B, H, S, D)  # device layout: MFMA k order + 16-byte chunk swizzle
    assert np.array_equal(got[:, :, :S], codes)
    assert np.all(got[:, :, S:] == 0)


def test_varlen_and_dispatchers(oracle, dev):
    import lowbit_quant_fa2_paddle_amd as lb
    H, D = 2, 64
    lens = [100, 256, 37]
    rng = np.random.default_rng(0)
    q = oracle.to_storage(rng.standard_normal((sum(lens), H, D), dtype=np.float32), "fp16")
    k = oracle.to_storage(rng.standard_normal((sum(lens), H, D), dtype=np.float32), "fp16")
    v = oracle.to_storage(rng.standard_normal((sum(lens), H, D), dtype=np.float32), "fp16")
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    tcu = torch.from_numpy(cu).to(dev)
    o = lb.lowbit_fa_varlen(tq, tk, tv, tcu, tcu, max(lens), max(lens), is_causal=True)
    assert tuple(o.shape) == (sum(lens), H, D)
    ref = oracle.lowbit_fa_varlen(q, k, v, cu, cu, is_causal=True, tail="neg_inf", amax_floor=1e-7)
    _o_close(_np(o), ref, "fp16")
    # auto dispatcher and the precision router
    q4, k4, v4 = oracle.make_inputs(1, 2, 128, 64, seed=1)
    t4 = [_t(a, "fp16", dev) for a in (q4, k4, v4)]
    o_a = lb.lowbit_fa_attn(*t4)
    o_b = lb.lowbit_fa_qk_int8_pv_fp16_triton(*t4)
    assert torch.equal(o_a, o_b)
    # precision router (src/core.py:1051-1096): N(0,1) -> max|x|/127 ~ 0.035 < 0.05 -> INT4; x3 -> INT8; x8 -> FP16
    assert lb.core.select_quantization(*t4) == "INT4"
    assert torch.equal(lb.lowbit_fa_multi_precision(*t4), lb.lowbit_fa_qk_int4_pv_fp16_triton(*t4))
    t12 = [t * 3 for t in t4]
    assert lb.core.select_quantization(*t12) == "INT8"
    assert torch.equal(lb.lowbit_fa_multi_precision(*t12), lb.lowbit_fa_qk_int8_pv_fp16_triton(*t12))
    t32 = [t * 8 for t in t4]
    assert lb.core.select_quantization(*t32) == "FP16"
    o_fp = lb.lowbit_fa_multi_precision(*t32)
    ref = oracle.sdpa_naive(*(x.astype(np.float64) * 8 for x in (q4, k4, v4)))
    # plain fp16 matmul SDPA on x8 inputs: scores ~ +-500 carry fp16 rounding of 0.25 -> loose check
    assert np.abs(_np(o_fp) - ref).max() <= 0.1 * np.abs(ref).max()


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_lazy_reference_rescale_branches(oracle, dev, D, causal):
    """The fp16-PV kernel exponentiates against a stale softmax reference and only moves it when a row sum
    blows up (attn_fwd16.hip, lazy softmax reference).  That branch is rare on random data, so force it: single keys aligned
    with single queries make one row's score jump by ~60 (finite overflow of the fp16 range) and by > 127
    (fp32 exp2 overflows to +inf) in LATER tiles, plus one spike in the very first tile.  Checked against the
    oracle on the FULL tensor."""
    import lowbit_quant_fa2_paddle_amd as lb
    S = 640
    q, k, v = oracle.make_inputs(1, 2, S, D, seed=77)
    kk = k.copy()
    f = 64.0 / D  # keep q.k of the spikes comparable across head dims
    kk[0, 0, 5] = oracle.to_storage(3.0 * f * q[0, 0, 600], "fp16")     # first tile, seen by row 600 (also causal)
    kk[0, 0, 300] = oracle.to_storage(6.0 * f * q[0, 0, 517], "fp16")   # tile 4: finite overflow for row 517
    kk[0, 0, 400] = oracle.to_storage(16.0 * f * q[0, 0, 433], "fp16")  # tile 6: +inf for row 433
    kk[0, 1, 130] = oracle.to_storage(-9.0 * f * q[0, 1, 200], "fp16")  # a hugely NEGATIVE score must not matter
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, kk, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, is_causal=causal, return_lse=True, smooth_k=False)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, kk, v, is_causal=causal, return_lse=True, smooth_k=False, amax_floor=1e-7)
    _o_close(_np(o), o_ref, "fp16")
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 1e-3 + 2.0 ** -20 * np.abs(lse_ref).max()
    # the spiked rows are (nearly) one-hot: they must reproduce the value row of their spike key
    assert np.abs(_np(o)[0, 0, 517] - v[0, 0, 300]).max() <= 2e-2
    assert np.abs(_np(o)[0, 0, 433] - v[0, 0, 400]).max() <= 2e-2


def test_quant_division_exceptional_divisor(oracle, dev):
    """The quantiser divides by the block scale with Markstein's reciprocal sequence, which is exact unless
    the scale's significand is all ones; that case takes the plain-division branch.  Force it: pick sm_scale so
    that max|x*sm_scale|/127 has an all-ones significand, and compare bit for bit with the oracle."""
    from lowbit_quant_fa2_paddle_amd import quant_per_block as qpb
    target = np.array([0x3C7FFFFF], dtype=np.uint32).view(np.float32)[0]  # ~0.0156, significand all ones
    sigma = None
    cand = np.float32(target * np.float32(127.0))
    for _ in range(400):
        if (np.float32(cand / np.float32(127.0))).view(np.uint32) == np.uint32(0x3C7FFFFF):
            sigma = cand
            break
        cand = np.nextafter(cand, np.float32(0), dtype=np.float32)
    if sigma is None:
        cand = np.float32(target * np.float32(127.0))
        for _ in range(400):
            if (np.float32(cand / np.float32(127.0))).view(np.uint32) == np.uint32(0x3C7FFFFF):
                sigma = cand
                break
            cand = np.nextafter(cand, np.float32(np.inf), dtype=np.float32)
    assert sigma is not None
    rng = np.random.default_rng(3)
    x = oracle.to_storage(rng.uniform(-1, 1, (1, 2, 256, 64)).astype(np.float32), "fp16")
    x[:, :, ::128, 0] = 1.0  # amax = 1.0 exactly in every 128-row block
    codes, scale = qpb.quantize(_t(x, "fp16", dev), sm_scale=float(sigma), qmax=127, blk=128)
    rc, rs = oracle.quant_per_block(x, sigma, 127.0, 128, amax_floor=1e-7)
    assert np.all(rs.view(np.uint32) == 0x3C7FFFFF), "test did not hit the exceptional divisor"
    assert np.array_equal(scale.cpu().numpy().view(np.uint32), rs.view(np.uint32))
    assert np.array_equal(codes.cpu().numpy(), rc)


def test_empty_inputs(dev):
    """Edge cases around emptiness: no queries -> empty output without a launch; no keys -> ValueError."""
    import lowbit_quant_fa2_paddle_amd as lb
    k = torch.randn(1, 2, 64, 64, device=dev).half()
    q0 = torch.empty(1, 2, 0, 64, device=dev, dtype=torch.float16)
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(q0, k, k, return_lse=True)
    assert tuple(o.shape) == (1, 2, 0, 64) and tuple(lse.shape) == (1, 2, 0)
    b0 = torch.empty(0, 2, 64, 64, device=dev, dtype=torch.float16)
    assert tuple(lb.lowbit_fa_qk_int8_pv_fp8_cuda(b0, b0, b0).shape) == (0, 2, 64, 64)
    k0 = torch.empty(1, 2, 0, 64, device=dev, dtype=torch.float16)
    with pytest.raises(ValueError, match="at least one key"):
        lb.lowbit_fa_qk_int8_pv_fp16_triton(k, k0, k0)


@pytest.mark.parametrize("layout,causal,pv", [("HND", False, "fp16"), ("NHD", True, "fp16"), ("HND", True, "fp8")])
def test_one_call_forward_equals_modular_entry_points(oracle, dev, layout, causal, pv):
    """lbfa_forward (what the operators call) == lbfa_mean_seq + 2 x lbfa_quant_per_block (+ lbfa_quant_v_fp8)
    + lbfa_attn_fwd composed on the host: O bit-identical, LSE (fix-up fused in the kernel epilogue vs torch ops) to 1e-6."""
    import lowbit_quant_fa2_paddle_amd as lb
    from lowbit_quant_fa2_paddle_amd import attn_qk_int8_per_block as attn, quant, quant_per_block as qpb
    B, H, Hkv, S, D = 2, 4, 2, 320, 128
    q, k, v = oracle.make_inputs(B, H, S, D, seed=41, layout=layout, Hkv=Hkv, k_bias=0.4)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    fn = lb.lowbit_fa_qk_int8_pv_fp16_triton if pv == "fp16" else lb.lowbit_fa_qk_int8_pv_fp8_cuda
    o, lse = fn(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True)
    sm_scale = D ** -0.5
    km = qpb.mean_seq(tk, layout)
    q8, qs, corr = qpb.quantize(tq, sm_scale=sm_scale * 1.44269504, qmax=127, blk=128, tensor_layout=layout, rowdot_vec=km)
    k8, ks = qpb.quantize(tk, sm_scale=1.0, qmax=127, blk=64, tensor_layout=layout, mean=km)
    if pv == "fp8":
        vin, vs, _ = quant.per_channel_fp8(tv, tensor_layout=layout)
    else:
        vin, vs = tv, None
    o2, lse2 = attn.forward(q8, k8, vin, qs, ks, tensor_layout=layout, output_dtype=torch.float16, return_lse=True,
                            is_causal=causal, v_scale=vs)
    assert torch.equal(o, o2)
    lse_host = lse2 / 1.44269504 + corr * sm_scale
    assert float((lse - lse_host).abs().max()) <= 1e-5


@pytest.mark.parametrize("D,layout,causal,pv", [
    (40, "HND", False, "fp16"), (72, "NHD", True, "fp16"), (96, "HND", True, "fp16"), (120, "NHD", False, "fp16"),
    (96, "HND", False, "fp8"), (48, "NHD", True, "fp8"),
    (36, "HND", False, "fp16"),   # not a multiple of 8: padded on the host as the reference does
])
def test_head_dims_padded_inside_the_kernels(oracle, dev, D, layout, causal, pv):
    """Head dims other than 64 / 128 (src/core.py:277-287 pads q, k, v with zeros and slices o): multiples of 8 are
    handled inside the kernels (padding channels are never read or written) and must give the padded result."""
    import lowbit_quant_fa2_paddle_amd as lb
    B, H, Hkv, S = 2, 4, 2, 200
    q, k, v = oracle.make_inputs(B, H, S, D, seed=17, layout=layout, Hkv=Hkv, k_bias=0.3)
    tq, tk, tv = (_t(a, "fp16", dev) for a in (q, k, v))
    fn = lb.lowbit_fa_qk_int8_pv_fp16_triton if pv == "fp16" else lb.lowbit_fa_qk_int8_pv_fp8_cuda
    o, lse = fn(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True)
    assert tuple(o.shape) == q.shape
    ro, rlse = oracle.lowbit_fa_forward(q, k, v, tensor_layout=layout, is_causal=causal, return_lse=True, pv=pv,
                                        tail="neg_inf", amax_floor=1e-7)
    if pv == "fp16":
        _o_close(_np(o), ro, "fp16")
    else:
        assert float(np.mean((_np(o) - ro) ** 2)) <= 1e-4
    assert np.abs(_np(lse) - rlse).max() <= 2e-3
    # identical to explicit zero padding on the host
    pad = (64 if D < 64 else 128) - D
    pq, pk, pv_ = (torch.nn.functional.pad(t, (0, pad)) for t in (tq, tk, tv))
    o_p = fn(pq, pk, pv_, tensor_layout=layout, is_causal=causal, sm_scale=D ** -0.5)[..., :D]
    assert torch.equal(o, o_p)


@pytest.mark.parametrize("pv", ["fp16", "fp8"])
def test_operator_is_hipgraph_capturable(oracle, dev, pv):
    """Every launch of the one-call path is asynchronous on the caller's stream (no host sync, no host read of device
    data), so the operator can be captured into a hipGraph (torch.cuda.CUDAGraph) and replayed on new inputs."""
    import lowbit_quant_fa2_paddle_amd as lb
    fn = lb.lowbit_fa_qk_int8_pv_fp16_triton if pv == "fp16" else lb.lowbit_fa_qk_int8_pv_fp8_cuda
    q, k, v = oracle.make_inputs(2, 4, 320, 64, seed=3, k_bias=0.2)
    sq, sk, sv = (_t(a, "fp16", dev) for a in (q, k, v))
    eager = fn(sq, sk, sv, is_causal=True, return_lse=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on the capture stream
        fn(sq, sk, sv, is_causal=True, return_lse=True)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(sq, sk, sv, is_causal=True, return_lse=True)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], eager[0]) and torch.equal(out[1], eager[1])
    # new data in the captured input buffers
    q2, k2, v2 = oracle.make_inputs(2, 4, 320, 64, seed=4, k_bias=-0.3)
    for dst, src in ((sq, q2), (sk, k2), (sv, v2)):
        dst.copy_(_t(src, "fp16", dev))
    g.replay()
    torch.cuda.synchronize()
    eager2 = fn(sq, sk, sv, is_causal=True, return_lse=True)
    assert torch.equal(out[0], eager2[0]) and torch.equal(out[1], eager2[1])


def test_c_abi_without_any_framework(dev):
    """examples/cabi_demo.cpp: a plain C++ host (hipMalloc + ONE lbfa_forward call, no torch / Python in the process)
    built against include/lowbit_fa.h, checked against exact fp32 attention on the host."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = os.path.join("/tmp", f"cabi_demo_{os.getpid()}")
    pkg = os.path.join(ROOT, "lowbit_quant_fa2_paddle_amd")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "examples", "cabi_demo.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + pkg, "-llowbit_fa_hip", "-Wl,-rpath," + pkg, "-o", exe],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("pv", ["fp16", "fp8", "unquantised"])
def test_packed_qkv_views_are_consumed_in_place(oracle, dev, pv):
    """q, k, v as strided views of one packed [B, S, 3, H, D] projection output (NHD, sequence stride 3*H*D): layouts are
    handed to the C ABI as strides (attn_qk_int8_per_block.py:183-196), no contiguous copies are made."""
    import lowbit_quant_fa2_paddle_amd as lb
    B, S, H, D = 2, 200, 4, 64
    q, k, v = oracle.make_inputs(B, H, S, D, seed=23, layout="NHD", k_bias=0.2)
    qkv = torch.stack([_t(a, "fp16", dev) for a in (q, k, v)], dim=2)  # [B, S, 3, H, D]
    tq, tk, tv = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    assert not tq.is_contiguous() and tq.stride(1) == 3 * H * D
    if pv == "unquantised":
        o = lb.core.flash_attn_fp16(tq, tk, tv, tensor_layout="NHD", is_causal=True)
        ref = np.transpose(oracle.sdpa_naive(*(np.transpose(a, (0, 2, 1, 3)).astype(np.float64) for a in (q, k, v)), is_causal=True), (0, 2, 1, 3))
        _o_close(_np(o), ref, "fp16")
        return
    fn = lb.lowbit_fa_qk_int8_pv_fp16_triton if pv == "fp16" else lb.lowbit_fa_qk_int8_pv_fp8_cuda
    o = fn(tq, tk, tv, tensor_layout="NHD", is_causal=True)
    o_c = fn(tq.contiguous(), tk.contiguous(), tv.contiguous(), tensor_layout="NHD", is_causal=True)
    assert torch.equal(o, o_c)
    # and the packed-batch operator on the flattened token axis of the same views
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=dev)
    flat = qkv.reshape(B * S, 3, H, D)
    ov = lb.lowbit_fa_varlen(flat[:, 0], flat[:, 1], flat[:, 2], cu, cu, S, S, is_causal=True)
    assert tuple(ov.shape) == (B * S, H, D) and torch.isfinite(ov).all()


@pytest.mark.parametrize("name", golden_names())
def test_operator_end_to_end_vs_reference_golden(oracle, dev, name):
    """fp16 / bf16 inputs -> public operator (one lbfa_forward call: in-kernel Q quantiser, K quantiser, attention) against
    the O that the reference's own quantiser + attention kernels produced for the same seeded inputs; and the in-kernel Q
    codes, observed through q_scale, against the reference quantiser's scales (bit-exact via the modular entry point)."""
    import lowbit_quant_fa2_paddle_amd as lb
    from lowbit_quant_fa2_paddle_amd import quant_per_block as qpb
    p, g = load_golden(name)
    q, k, v = golden_inputs(oracle, p)
    tq, tk, tv = (_t(a, p["dtype"], dev) for a in (q, k, v))
    kw = dict(tensor_layout=p["layout"], is_causal=p["causal"], smooth_k=p["smooth_k"])
    if p["q_qmax"] == 127 and p["k_qmax"] == 127:
        o = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, **kw)
    else:
        o = lb.lowbit_fa_qk_int4_pv_fp16_triton(tq, tk, tv, q_bits=4 if p["q_qmax"] == 7 else 8, **kw)
    _golden_o_close(_np(o), g["o"], v, p)
    # the Q quantiser on the padded tensor, with the scale factor formed as the reference's Python does (in double)
    D = p["D"]
    Dp = 64 if D <= 64 else 128
    tqp = torch.nn.functional.pad(tq, (0, Dp - D)) if Dp != D else tq
    q8, qs = qpb.quantize(tqp, sm_scale=D ** -0.5 * 1.44269504, qmax=p["q_qmax"], blk=128, tensor_layout=p["layout"])
    assert np.array_equal(q8.cpu().numpy(), g["q_i8"])
    assert np.array_equal(qs.cpu().numpy().view(np.uint32), g["q_scale"].view(np.uint32))
