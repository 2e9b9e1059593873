"""The oracle's e4m3fn helpers (`e4m3fn_round`, `e4m3fn_encode`, `e4m3fn_decode`: what the fp8-PV restatement rounds P and V
with, csrc/numeric_conversion.cuh:39-54 `cvt.rn.satfinite.e4m3x2.f32`) against an independent implementation of the OCP e4m3fn
format that IS available in this image: torch.float8_e4m3fn.  The fp8-PV path itself stays "parity unpinned" (the reference's
CUDA kernels cannot run here and hold no fixture) - this pins only the number format under it."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
if not hasattr(torch, "float8_e4m3fn"):
    pytest.skip("torch.float8_e4m3fn not available", allow_module_level=True)


def _torch_round(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.float8_e4m3fn).to(torch.float32).numpy()


def _torch_codes(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def _grid():
    """every finite e4m3fn value, decoded by torch from the 256 bit patterns (0x7f / 0xff are the NaNs)"""
    codes = np.arange(256, dtype=np.uint8)
    vals = torch.from_numpy(codes).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    ok = np.isfinite(vals)
    return codes[ok], vals[ok]


def test_decode_and_encode_all_codes(oracle):
    codes, vals = _grid()
    assert len(codes) == 254 and vals.max() == 448.0 and vals.min() == -448.0
    assert np.array_equal(oracle.e4m3fn_decode(codes), vals)
    enc = oracle.e4m3fn_encode(vals)
    # +0 / -0 keep their sign bit
    assert np.array_equal(enc, codes), np.argwhere(enc != codes)[:5]
    assert np.array_equal(oracle.e4m3fn_round(vals), vals)  # grid points are fixed points


def test_round_midpoints_ties_to_even_and_neighbours(oracle):
    _, vals = _grid()
    pos = np.sort(vals[vals >= 0])  # 0, 2^-9, ..., 448
    mid = ((pos[:-1].astype(np.float64) + pos[1:].astype(np.float64)) / 2).astype(np.float32)  # exactly representable in fp32
    below, above = np.nextafter(mid, np.float32(0)), np.nextafter(mid, np.float32(np.inf))
    for x in (mid, below, above, -mid, -below, -above):
        want = _torch_round(x)
        got = oracle.e4m3fn_round(x)
        assert np.array_equal(got, want), (x[got != want][:5], got[got != want][:5], want[got != want][:5])
        assert np.array_equal(oracle.e4m3fn_encode(got), _torch_codes(x))


def test_round_dense_sample(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.uniform(-448, 448, 200001).astype(np.float32),
        (rng.standard_normal(100000) * 2.0 ** rng.integers(-12, 9, 100000)).astype(np.float32),
        np.float32(2.0) ** np.arange(-20, 9, dtype=np.float32), np.zeros(1, np.float32), -np.zeros(1, np.float32),
    ])
    x = x[np.abs(x) <= 448.0]
    assert np.array_equal(oracle.e4m3fn_round(x), _torch_round(x))
    assert np.array_equal(oracle.e4m3fn_encode(oracle.e4m3fn_round(x)), _torch_codes(x))


def test_saturation(oracle):
    """satfinite: everything beyond 448 becomes +-448 (the torch cast turns values past the last rounding boundary, 464, into
    NaN instead - the reference's PTX conversion saturates, csrc/numeric_conversion.cuh:39-54).  Up to the boundary both agree."""
    x = np.array([448.0, 449.0, 463.9, 464.0], np.float32)
    assert np.array_equal(oracle.e4m3fn_round(x), np.full(4, 448.0, np.float32))
    assert np.array_equal(_torch_round(x[:3]), np.full(3, 448.0, np.float32))
    big = np.array([464.5, 1e3, 65504.0, 3e38, np.inf], np.float32)
    for sgn in (1.0, -1.0):
        assert np.array_equal(oracle.e4m3fn_round(sgn * big), np.full(len(big), sgn * 448.0, np.float32))
        assert np.array_equal(oracle.e4m3fn_encode(oracle.e4m3fn_round(sgn * big)), np.full(len(big), 0x7e | (0x80 if sgn < 0 else 0), np.uint8))
