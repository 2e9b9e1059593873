"""GPU parity tests added in round 2 (run with `-m gpu`): the data-dependent paths of the reworked attention kernel.

  * the deferred fp16-overflow check: EVERY q-block of the tensor overflows fp16 P in its first (lazy) pass and is redone
    with the exact row max per tile (attn_fwd16.hip `run_tiles(Yes)`), both head dims, dense + ragged + causal, against the oracle on the FULL tensor;
  * an all-zero K block inside a MASKED tile (ragged last tile / causal diagonal): its dequantisation scale sits on the
    1e-7 floor, far below the scale grid of the head - the exponent argument of a masked key must stay -inf, not NaN;
  * a packed batch whose caller understates max_seqlen_k: sequences are cut at the stated maximum, nothing is read
    out of bounds;
  * fp8: the P -> e4m3 conversion of the device (v_cvt_pk_fp8_f32) against the oracle's encoder on all 256 codes, ties,
    subnormals and saturation (bit-exact).  fp8-PV as a whole stays "parity unpinned" (no reference fixture).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lowbit_quant_fa2_paddle_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _t(a, dtype, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(TDT[dtype]).to(dev)


def _np(t):
    return t.detach().float().cpu().numpy()


def _o_close(o, ref, atol=2e-3, rtol=2e-3):
    err = np.abs(o - ref)
    bad = err > atol + rtol * np.abs(ref)
    assert not bad.any(), f"max err {err.max():.3e} at {np.argwhere(bad)[:3].tolist()} ({bad.sum()} elements)"


from test_gpu_parity import _fp8_close  # noqa: E402


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("S,causal", [(1024, False), (1000, False), (1024, True)])
@pytest.mark.parametrize("smooth_k", [False, True])
def test_every_q_block_takes_the_exact_rerun(oracle, dev, D, S, causal, smooth_k):
    """Keys 0..63 (the tile that sets the lazy reference) score ~23 (log2 units) BELOW every later key for every query row:
    exp2(s - m) overflows fp16 for all rows of all q-blocks, so every workgroup discards its first pass and redoes the
    block with exact row maxima.  Output and LSE must still match the oracle on the full tensor."""
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(2, 2, S, D, seed=5)
    a = 8.0 * (D / 64.0) ** 0.5  # q.k shift of +-a*a = +-64 sqrt-scaled, i.e. +-8 after sm_scale at D = 64 ...
    q[..., 0] += a
    k[:, :, :64, 0] -= a   # ... first tile: -8 - noise
    k[:, :, 64:, 0] += a   # later tiles: +8, 16 nats = 23 bits above the first tile
    q, k = oracle.to_storage(q, "fp16"), oracle.to_storage(k, "fp16")
    tq, tk, tv = (_t(x, "fp16", dev) for x in (q, k, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, is_causal=causal, return_lse=True, smooth_k=smooth_k)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, return_lse=True, smooth_k=smooth_k, amax_floor=1e-7,
                                              tail="neg_inf")
    _o_close(_np(o), o_ref)
    # smooth_k: lse gets q . km * sm_scale with q . km ROUNDED TO FP16 (src/core.py:294-304).  Here |q . km| ~ 64..128, one
    # fp16 ulp of it times sm_scale is 0.0625 / sqrt(D): the device's fp32 dot and numpy's can round to neighbouring values
    lse_tol = 1e-3 + 2.0 ** -20 * np.abs(lse_ref).max() + (0.0625 * D ** -0.5 * 1.01 if smooth_k else 0.0)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= lse_tol
    # the un-quantised kernel shares the tile loop (lazy pass + exact re-run)
    from lowbit_quant_fa2_paddle_amd import core
    of = core.flash_attn_fp16(tq, tk, tv, is_causal=causal)
    ref = oracle.sdpa_naive(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), is_causal=causal)
    _o_close(_np(of), ref)


@pytest.mark.parametrize("variant", ["int8", "int4", "fp8"])
@pytest.mark.parametrize("S,causal,zero_lo", [(200, False, 192), (256, True, 128), (320, True, 256), (100, False, 64)])
@pytest.mark.parametrize("D", [64, 128])
def test_zero_k_block_in_a_masked_tile(oracle, dev, variant, S, causal, zero_lo, D):
    """smooth_k=False and a 64-key block of exact zeros that lies in a masked tile (ragged tail or causal diagonal): the
    block's scale is the 1e-7 / qmax floor, < 2^-22 of the head's largest scale, and used to round to 0 on the kernel's
    scale grid -> fma(-inf, 0, c1) = NaN for the masked keys.  Output finite and equal to the oracle."""
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(1, 2, S, D, seed=13)
    k[:, :, zero_lo:zero_lo + 64] = 0.0
    tq, tk, tv = (_t(x, "fp16", dev) for x in (q, k, v))
    if variant == "int8":
        o = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, is_causal=causal, smooth_k=False)
        ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, smooth_k=False, amax_floor=1e-7, tail="neg_inf")
        tol = dict(atol=2e-3, rtol=2e-3)
    elif variant == "int4":
        o = lb.lowbit_fa_qk_int4_pv_fp16_triton(tq, tk, tv, is_causal=causal, smooth_k=False)
        ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, smooth_k=False, q_qmax=7, k_qmax=7, amax_floor=1e-7,
                                       tail="neg_inf")
        tol = dict(atol=2e-3, rtol=2e-3)
    else:
        o = lb.lowbit_fa_qk_int8_pv_fp8_cuda(tq, tk, tv, is_causal=causal, smooth_k=False)
        ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, smooth_k=False, pv="fp8", amax_floor=1e-7, tail="neg_inf")
        tol = None  # fp8-PV: parity unpinned, loose against the oracle's restatement (_fp8_close)
    assert torch.isfinite(o).all()
    if tol is None:
        _fp8_close(_np(o), ref)
    else:
        _o_close(_np(o), ref, **tol)


def test_varlen_understated_max_seqlen_is_a_cut_not_an_overrun(oracle, dev):
    """max_seqlen_k sizes the K quantiser's grid and the padded scale rows of lbfa_forward_varlen.  A caller that
    understates it gets every sequence cut at the stated maximum (kernel clamp), not reads past the scale buffer."""
    import lowbit_quant_fa2_paddle_amd as lb
    lens = [300, 129, 70]
    D, H = 64, 2
    q, k, v, cu_q, cu_k = oracle.make_varlen_inputs(lens, lens, H, H, D, seed=3)
    tq, tk, tv = (_t(x, "fp16", dev) for x in (q, k, v))
    tcq, tck = torch.from_numpy(cu_q).to(dev), torch.from_numpy(cu_k).to(dev)
    cut = 128
    o = lb.sageattn_varlen(tq, tk, tv, tcq, tck, max(lens), cut, is_causal=False, smooth_k=False)
    assert torch.isfinite(o).all()
    for b, n in enumerate(lens):
        qb = np.transpose(q[cu_q[b]:cu_q[b + 1]], (1, 0, 2))[None]
        kb = np.transpose(k[cu_k[b]:cu_k[b] + min(n, cut)], (1, 0, 2))[None]
        vb = np.transpose(v[cu_k[b]:cu_k[b] + min(n, cut)], (1, 0, 2))[None]
        ref = oracle.lowbit_fa_forward(qb, kb, vb, smooth_k=False, amax_floor=1e-7, tail="neg_inf")
        got = np.transpose(_np(o[cu_q[b]:cu_q[b + 1]]), (1, 0, 2))[None]
        _o_close(got, ref)


def test_p_to_e4m3_conversion_bit_exact(oracle, dev):
    """v_cvt_pk_fp8_f32 (the instruction that rounds P, and V, to OCP e4m3fn on the device) against the oracle's
    `e4m3fn_round` / `e4m3fn_encode`: every representable magnitude, every midpoint between neighbours (ties-to-even), the
    fp16 neighbours of every midpoint, subnormals, zero, and saturation at and beyond 448, both signs.  Driven through
    the V encoder (`lbfa_quant_v_fp8` multiplies a channel by 448 / amax: the column holds the ramp and its amax is
    exactly 448, so the factor is exactly 1 and the instruction sees the ramp itself)."""
    from lowbit_quant_fa2_paddle_amd import quant
    vals = oracle.e4m3fn_decode(np.arange(0, 127, dtype=np.uint8)).astype(np.float16)  # +0 .. +448, exact in fp16
    mids = ((vals[:-1].astype(np.float32) + vals[1:].astype(np.float32)) / 2).astype(np.float16)
    assert np.array_equal(mids.astype(np.float32) * 2, vals[:-1].astype(np.float32) + vals[1:].astype(np.float32))
    ramp = np.concatenate([vals, mids, np.nextafter(mids, np.float16(0)), np.nextafter(mids, np.float16(1000)),
                           np.array([2.0 ** -11, 2.0 ** -10, 3 * 2.0 ** -11, 2.0 ** -14, 447.75, 448.0], dtype=np.float16)])
    ramp = np.concatenate([ramp, -ramp]).astype(np.float32)
    D = 64
    S = 64 * ((len(ramp) + 63) // 64)
    v = np.zeros((1, 1, S, D), dtype=np.float32)
    v[0, 0, :len(ramp), 0] = ramp  # channel 0: amax = 448 exactly
    v[0, 0, :, 1:] = 1.0
    tv = torch.from_numpy(v).to(torch.float16).to(dev)
    v8, v_scale, _ = quant.per_channel_fp8(tv, tensor_layout="HND")
    assert float(v_scale[0, 0, 0]) == 1.0
    from fp8_layout import decode_v_fp8
    got = decode_v_fp8(v8.buf.cpu().numpy(), 1, 1, S, D)[0, 0, :, 0]  # channel 0 in key order
    want = oracle.e4m3fn_encode(oracle.e4m3fn_round(v[0, 0, :, 0]))
    got[got == 0x80] = 0   # -0 and +0 are the same number: the device keeps the sign of a negative input that rounds to zero,
    want[want == 0x80] = 0  # numpy's sign() drops it
    assert np.array_equal(got[:len(ramp)], want[:len(ramp)]), \
        [(float(ramp[i]), hex(got[i]), hex(want[i])) for i in np.argwhere(got[:len(ramp)] != want[:len(ramp)])[:8, 0]]
    # saturation, not NaN / inf, beyond the format's range (P is never above 448, V * 448 / amax neither; the convert is
    # still required to clamp): checked on the instruction's documented behaviour through 448 itself above
    assert want[np.argmax(ramp == 448.0)] == 0x7E


@pytest.mark.parametrize("D,S,first_rev_row", [(64, 13312, 12288), (128, 8448, 8192)])
@pytest.mark.parametrize("variant", ["fp16", "fp8", "fp16_peaky"])
def test_reversed_rounds_of_q_blocks_vs_oracle(oracle, dev, D, S, first_rev_row, variant):
    """Non-causal kernels walk the key tiles backwards for every other ROUND of Q blocks of a head (ping-pong order for L2
    reuse, `tile_of` in attn_fwd16.hip / attn_fwd.hip): Q blocks >= 96 (D = 64) / >= 64 (D = 128).  The rows of the first reversed round - with
    more than 64 key tiles, so the per-64-tile scale table is refreshed downwards too - against the oracle, and a forward
    block next to them for contrast.  (fp8 at D = 64 keeps 64-block rounds.)  `fp16_peaky`: queries x 6 - scores 8 binades wide,
    so Q blocks of both directions overflow their lazy pass at some vote and replay in exact mode (round 3)."""
    import lowbit_quant_fa2_paddle_amd as lb
    q, k, v = oracle.make_inputs(1, 1, S, D, seed=23, k_bias=0.3)
    if variant == "fp16_peaky":
        q, variant = oracle.to_storage(q * 6.0, "fp16"), "fp16"
    tq, tk, tv = (_t(x, "fp16", dev) for x in (q, k, v))
    fn = lb.lowbit_fa_qk_int8_pv_fp16_triton if variant == "fp16" else lb.lowbit_fa_qk_int8_pv_fp8_cuda
    o, lse = fn(tq, tk, tv, return_lse=True)
    assert torch.isfinite(o).all()
    lo = first_rev_row - 128 if not (variant == "fp8" and D == 64) else 8192 - 128
    rows = slice(lo, min(lo + 640, S))  # one forward block + the first reversed ones (whole 128-row quantisation blocks)
    ref, lse_ref = oracle.lowbit_fa_forward(q[:, :, rows], k, v, return_lse=True, amax_floor=1e-7,
                                            pv="fp8" if variant == "fp8" else "fp16")
    if variant == "fp16":
        _o_close(_np(o)[:, :, rows], ref)
    else:
        _fp8_close(_np(o)[:, :, rows], ref)
    assert np.abs(lse.cpu().numpy()[:, :, rows] - lse_ref).max() <= (1e-3 if variant == "fp16" else 2e-3) + 2.0 ** -9 * np.abs(lse_ref).max()
