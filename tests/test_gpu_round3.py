"""GPU parity tests added in round 3 (run with `-m gpu`).

  * the lazy softmax reference with EARLY overflow votes and a partial replay (attn_fwd16.hip, "tile loop"): the first fp16-P
    overflow of a Q block is placed in key tile 1, 3 or 40 of 64 - before, between and after the geometric vote points - for
    all rows, for the rows of one wave only (the other three waves keep their accumulators and skip the replay), or for a single
    row; and "spike" inputs whose later tiles fall 2^36 below the reference, so that the exact path skips them as dead;
  * the reference's own benchmark distribution q, k = randint(-100, 100) (utils/benchmark.py:215-230): one-hot softmax rows,
    every Q block on the exact path, against the oracle;
  * the LSE of rows whose softmax has one dominant key (one-hot rows, S = 1, the first rows of a causal block): the kernel sums
    fp16-ROUNDED P (the row sums ride on the all-ones MFMA block), the reference sums fp32 P before rounding
    (src/triton/attn_qk_int8_per_block.py:54-60): |dLSE| <= log2(1 + 2^-11) * ln 2 ~ 4.9e-4 nat, asserted with its bound.
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


def _shifted_inputs(oracle, S, D, first_high_tile, rows, spike, seed=21):
    """N(0,1) inputs whose channel 0 carries a step: keys of tiles >= first_high_tile score 2 a^2 sm_scale log2(e) ~ 23 (36
    for `spike`) binades above the earlier ones for the chosen query rows; `spike`: only that one tile is high."""
    q, k, v = oracle.make_inputs(1, 2, S, D, seed=seed)
    a = (10.0 if spike else 8.0) * (D / 64.0) ** 0.5
    r = np.arange(S) % 128
    sel = {"all": r >= 0, "one_wave": (r >= 32) & (r < 64), "one_row": r == 77}[rows]
    q[:, :, sel, 0] += a
    lo = 64 * first_high_tile
    k[:, :, :lo, 0] -= a
    k[:, :, lo:, 0] += a
    if spike:
        k[:, :, lo + 64:, 0] -= 2 * a
    return oracle.to_storage(q, "fp16"), oracle.to_storage(k, "fp16"), v


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("rows", ["all", "one_wave", "one_row"])
@pytest.mark.parametrize("first_high_tile", [1, 3, 40])
@pytest.mark.parametrize("spike", [False, True])
def test_overflow_first_appears_in_tile_k(oracle, dev, D, rows, first_high_tile, spike):
    import lowbit_quant_fa2_paddle_amd as lb
    S = 4096
    q, k, v = _shifted_inputs(oracle, S, D, first_high_tile, rows, spike)
    tq, tk, tv = (_t(x, "fp16", dev) for x in (q, k, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, return_lse=True, smooth_k=False)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, k, v, return_lse=True, smooth_k=False, amax_floor=1e-7, tail="neg_inf")
    _o_close(_np(o), o_ref)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 1e-3 + 2.0 ** -20 * np.abs(lse_ref).max()


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("first_high_tile", [2, 5])
def test_overflow_votes_causal_and_ragged(oracle, dev, causal, first_high_tile):
    """The same step inputs through the causal instance (Q blocks of every length: 2 .. 16 main tiles + the diagonal) and, non-causal,
    with a ragged key tail (the masked last tile after an overflowed lazy pass) and bf16 storage."""
    import lowbit_quant_fa2_paddle_amd as lb
    S, D = (1024, 64) if causal else (1000, 128)
    q, k, v = _shifted_inputs(oracle, S if causal else 1024, D, first_high_tile, "one_wave", False, seed=8)
    q, k, v = q[:, :, :S], k[:, :, :S], v[:, :, :S]
    for dt in ("fp16", "bf16"):
        qs, ks, vs = (oracle.to_storage(x, dt) for x in (q, k, v))
        tq, tk, tv = (_t(x, dt, dev) for x in (qs, ks, vs))
        o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, is_causal=causal, return_lse=True, smooth_k=True)
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        o_ref, lse_ref = oracle.lowbit_fa_forward(qs, ks, vs, dtype=dt, is_causal=causal, return_lse=True, smooth_k=True,
                                                  amax_floor=1e-7, tail="neg_inf")
        ulp = 2.0 ** -8 if dt == "bf16" else 0.0  # bf16 output: one ulp on top
        _o_close(_np(o), o_ref, atol=2e-3, rtol=2e-3 + ulp)


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_reference_bench_distribution_randint(oracle, dev, D, causal):
    """q, k = randint(-100, 100), v ~ N(0,1) (utils/benchmark.py:215-230): scores are thousands of binades apart, every row is
    one-hot, every Q block leaves the lazy pass at the first vote."""
    import lowbit_quant_fa2_paddle_amd as lb
    from lowbit_quant_fa2_paddle_amd import core
    q, k, v = oracle.make_inputs(1, 2, 1024, D, seed=3, dist="randint")
    tq, tk, tv = (_t(x, "fp16", dev) for x in (q, k, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, is_causal=causal, return_lse=True)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, return_lse=True, amax_floor=1e-7, tail="neg_inf")
    _o_close(_np(o), o_ref)
    # |lse| ~ 1e4 here: the tolerance scales with it (fp32 ulp of the value, and of q . km rounded to fp16 - src/core.py:294-304)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 1e-3 + 2.0 ** -10 * np.abs(lse_ref).max()
    o4 = lb.lowbit_fa_qk_int4_pv_fp16_triton(tq, tk, tv, is_causal=causal)
    o4_ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, q_qmax=7, k_qmax=7, amax_floor=1e-7, tail="neg_inf")
    _o_close(_np(o4), o4_ref)
    of = core.flash_attn_fp16(tq, tk, tv, is_causal=causal)
    ref = oracle.sdpa_naive(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), is_causal=causal)
    _o_close(_np(of), ref)


LSE_BOUND = float(np.log(1.0 + 2.0 ** -11))  # 4.88e-4 nat: one fp16 rounding of the dominant P


@pytest.mark.parametrize("case", ["one_hot_tile0", "one_hot_late", "single_key", "causal_first_rows", "two_equal"])
@pytest.mark.parametrize("D", [64, 128])
def test_lse_of_dominant_key_rows_within_the_fp16_p_bound(oracle, dev, case, D):
    """Row sums over fp16-rounded P (divergence list): for a row with ONE contributing key the row sum is that
    key's P rounded to fp16, relative error <= 2^-11 -> |dLSE| <= ln(1 + 2^-11); it falls with the number of contributing
    keys (independent roundings).  Checked on its worst cases against the oracle (fp32 sums before rounding): a dominant key
    ~25 nats above all others in the tile that sets the lazy reference (lazy path, P = 2^-delta off the fp16 grid) or in a
    later tile (overflow -> exact path), two equal dominant keys, a single key, and the first rows of a causal block."""
    import lowbit_quant_fa2_paddle_amd as lb
    causal = case == "causal_first_rows"
    S = 1 if case == "single_key" else 256
    q, k, v = oracle.make_inputs(1, 2, S, D, seed=17)
    if case in ("one_hot_tile0", "one_hot_late", "two_equal"):
        rng = np.random.default_rng(5)
        u = rng.standard_normal((1, 2, 1, D)).astype(np.float32)
        u /= np.linalg.norm(u, axis=-1, keepdims=True)
        idx = 10 if case == "one_hot_tile0" else 100
        k[:, :, idx:idx + 1] = 2.0 * D ** 0.5 * u          # q'.k / sqrt(D) = 2 (c + N(0,1)) with q' = q + c u
        q[:] = q + 16.0 * u
        if case == "two_equal":
            k[:, :, 200] = k[:, :, idx]
    q, k = oracle.to_storage(q, "fp16"), oracle.to_storage(k, "fp16")
    tq, tk, tv = (_t(x, "fp16", dev) for x in (q, k, v))
    o, lse = lb.lowbit_fa_qk_int8_pv_fp16_triton(tq, tk, tv, is_causal=causal, return_lse=True, smooth_k=False)
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, k, v, is_causal=causal, return_lse=True, smooth_k=False, amax_floor=1e-7,
                                              tail="neg_inf")
    _o_close(_np(o), o_ref)
    err = np.abs(lse.cpu().numpy() - lse_ref)
    # + the fp32 arithmetic of the fix-up itself (log2, one multiply: ulps of |lse|)
    assert err.max() <= LSE_BOUND + 2.0 ** -21 * np.abs(lse_ref).max(), f"max |dLSE| {err.max():.3e}"


N_PEAKY = int(__import__("os").environ.get("LBFA_PEAKY_N", "24"))  # one-off hunts: LBFA_PEAKY_N=300


@pytest.mark.parametrize("seed", range(N_PEAKY))
def test_fuzz_peaky_inputs(oracle, dev, seed):
    """Seeded fuzz of the overflow machinery: random shapes with 4..32 key tiles, queries scaled by 3..8 (scores 4..12 binades
    wide: some rows of some Q blocks overflow the lazy pass at some vote, others never), or the randint distribution;
    int8 / int4-range codes, causal or not, both layouts, both dtypes, with the LSE."""
    import lowbit_quant_fa2_paddle_amd as lb
    rng = np.random.default_rng(7000 + seed)
    D = int(rng.choice([64, 128, 80]))
    S = int(rng.choice([256, 320, 500, 768, 1000, 1536, 2048]))
    causal = bool(rng.integers(0, 2))
    layout = str(rng.choice(["HND", "NHD"]))
    dt = str(rng.choice(["fp16", "bf16"]))
    randint = seed % 4 == 3
    Hkv = int(rng.choice([1, 2]))
    H = Hkv * int(rng.choice([1, 2]))
    q, k, v = oracle.make_inputs(1, H, S, D, seed=seed, layout=layout, dtype=dt, Hkv=Hkv, dist="randint" if randint else "normal",
                                 k_bias=float(rng.choice([0.0, 0.4])))
    if not randint:
        q = oracle.to_storage(q * float(rng.choice([3.0, 5.0, 8.0])), dt)
    tq, tk, tv = (_t(a, dt, dev) for a in (q, k, v))
    int4 = seed % 3 == 2
    fn = lb.lowbit_fa_qk_int4_pv_fp16_triton if int4 else lb.lowbit_fa_qk_int8_pv_fp16_triton
    o, lse = fn(tq, tk, tv, tensor_layout=layout, is_causal=causal, return_lse=True)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    qm = dict(q_qmax=7, k_qmax=7) if int4 else {}
    o_ref, lse_ref = oracle.lowbit_fa_forward(q, k, v, dtype=dt, tensor_layout=layout, is_causal=causal, return_lse=True,
                                              amax_floor=1e-7, tail="neg_inf", **qm)
    _o_close(_np(o), o_ref, rtol=2e-3 + (2.0 ** -7 if dt == "bf16" else 0.0))  # bf16: one ulp of the output on top
    # |lse| can reach 1e4 (randint): fp32 ulps of the value and of q . km rounded to the storage dtype (src/core.py:294-304)
    ulp = 2.0 ** -10 if dt == "fp16" else 2.0 ** -7
    assert np.abs(_np(lse) - lse_ref).max() <= 1e-3 + ulp * np.abs(lse_ref).max()
