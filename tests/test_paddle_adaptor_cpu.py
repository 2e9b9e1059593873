"""`PaddleOps` (lowbit_quant_fa2_paddle_amd/_tensor.py) driven once on CPU.  Paddle is not installed in the build image
(and nothing may be installed), so a STAND-IN module object is registered as `paddle` for the duration of this test only:
numpy-backed tensors that expose what the adaptor touches (`.dtype .shape .strides .place .data_ptr() .astype()
.contiguous() .to()`, `paddle.empty / zeros / concat / cumsum`, `paddle.device.set_device / current_stream`).  It checks
the adaptor's own logic - the five primitives, head-dim padding for dense AND packed tensors, int32 cu_seqlens, the
cumulative scale offsets, the device string - not Paddle."""
import sys
import types

import numpy as np
import pytest


class _Place:
    def __init__(self, gpu_id=None):
        self._id = gpu_id

    def is_gpu_place(self):
        return self._id is not None

    def is_cpu_place(self):
        return self._id is None

    def get_device_id(self):
        return self._id

    def __str__(self):
        return "Place(cpu)" if self._id is None else f"Place(gpu:{self._id})"


class _Tensor:
    __module__ = "paddle.stand_in"

    def __init__(self, a, place=None):
        self.a = np.asarray(a)
        self.place = place or _Place(0)

    dtype = property(lambda self: self.a.dtype.name)
    shape = property(lambda self: list(self.a.shape))
    ndim = property(lambda self: self.a.ndim)
    strides = property(lambda self: [s // self.a.itemsize for s in self.a.strides])  # Paddle: element strides

    def data_ptr(self):
        return self.a.ctypes.data

    def astype(self, dt):
        return _Tensor(self.a.astype(dt), self.place)

    def contiguous(self):
        return _Tensor(np.ascontiguousarray(self.a), self.place)

    def to(self, place):
        return _Tensor(self.a, place)


_Tensor.__module__ = "paddle.stand_in"


@pytest.fixture()
def fake_paddle(monkeypatch):
    m = types.ModuleType("paddle")
    for n in ("int8", "float32", "uint8", "float16", "int32", "int64"):
        setattr(m, n, n)
    m.bfloat16 = "bfloat16"
    # like Paddle: allocations land on the device selected by the last set_device call
    m.empty = lambda shape, dtype: _Tensor(np.empty(shape, dtype=dtype), _Place(int(calls[-1].split(":")[1]) if calls and calls[-1] != "cpu" else None))
    m.zeros = lambda shape, dtype: _Tensor(np.zeros(shape, dtype=dtype))
    m.concat = lambda ts, axis=0: _Tensor(np.concatenate([t.a for t in ts], axis=axis), ts[0].place)
    m.cumsum = lambda t, axis=0: _Tensor(np.cumsum(t.a, axis=axis), t.place)
    calls = []
    stream = types.SimpleNamespace(stream_base=types.SimpleNamespace(raw_stream=0x1234))
    m.device = types.SimpleNamespace(set_device=lambda s: calls.append(s), current_stream=lambda: stream)
    monkeypatch.setitem(sys.modules, "paddle", m)
    from lowbit_quant_fa2_paddle_amd import _tensor
    monkeypatch.setattr(_tensor, "_OPS", {})
    return m, calls


def test_paddle_ops_primitives(fake_paddle):
    m, calls = fake_paddle
    from lowbit_quant_fa2_paddle_amd import _lib, _tensor
    from lowbit_quant_fa2_paddle_amd.quant_per_block import _bhs
    x = _Tensor(np.zeros((2, 3, 5, 40), dtype=np.float16))
    ops = _tensor.ops_for(x)
    assert ops.name == "paddle" and isinstance(ops, _tensor.PaddleOps)
    assert ops.dtype_code(x) == _lib.LBFA_F16 and ops.dtype_code(_Tensor(np.zeros(2, np.float32))) is None
    assert ops.ptr(x) == x.a.ctypes.data
    assert ops.shape(x) == (2, 3, 5, 40) and ops.strides(x) == (600, 200, 40, 1)
    nhd = _Tensor(np.zeros((2, 5, 3, 40), dtype=np.float16))
    assert _bhs(ops.shape(nhd), ops.strides(nhd), "NHD") == ((2, 3, 5), (600, 40, 120))
    assert ops.is_gpu(x) and not ops.is_gpu(_Tensor(np.zeros(1), _Place(None)))
    assert ops.same_device(x, nhd) and not ops.same_device(x, _Tensor(np.zeros(1), _Place(1)))
    y = _Tensor(np.zeros((2, 2), dtype=np.float16), _Place(3))
    e = ops.empty((4, 2), ops.float32, y)  # selects y's device, then allocates there: no `.to(place)` copy
    assert e.shape == [4, 2] and e.dtype == "float32" and str(e.place) == str(y.place) and calls == ["gpu:3"]
    assert ops.stream(x) == 0x1234
    with ops.device_guard(x):
        pass
    assert calls == ["gpu:3", "gpu:0"]


def test_paddle_ops_pad_last_dense_and_packed(fake_paddle):
    from lowbit_quant_fa2_paddle_amd import _tensor
    rng = np.random.default_rng(0)
    for shape in [(2, 3, 5, 40), (7, 3, 40), (4, 40)]:  # dense [B,H,S,D], packed varlen [T,H,D], and a 2-D corner
        a = rng.standard_normal(shape).astype(np.float16)
        t = _Tensor(a)
        ops = _tensor.ops_for(t)
        p = ops.pad_last(t, 24)
        assert p.shape == list(shape[:-1]) + [64]
        assert np.array_equal(p.a[..., :40], a) and not p.a[..., 40:].any()
        assert p.dtype == "float16"


def test_paddle_ops_cu_seqlens_helpers(fake_paddle):
    from lowbit_quant_fa2_paddle_amd import _tensor
    cu = _Tensor(np.array([0, 5, 12], dtype=np.int64)[::1])
    ops = _tensor.ops_for(cu)
    c32 = ops.as_int32(cu)
    assert c32.dtype == "int32" and c32.a.flags["C_CONTIGUOUS"] and c32.a.tolist() == [0, 5, 12]
    nblk = _Tensor(np.array([2, 1, 4], dtype=np.int64))
    off = ops.cumsum0_pad(nblk)
    assert off.dtype == "int32" and off.a.tolist() == [0, 2, 3, 7]
    cat = ops.cat0([_Tensor(np.ones((2, 3), np.float16)), _Tensor(np.zeros((1, 3), np.float16))])
    assert cat.shape == [3, 3]


def test_device2str():
    from lowbit_quant_fa2_paddle_amd._tensor import device2str
    assert device2str(3) == "gpu:3"
    assert device2str("cuda:1") == "gpu:1" and device2str("cuda", 2) == "gpu:2" and device2str("gpu:0") == "gpu:0"
    assert device2str("cpu") == "cpu" and device2str(None) == "cpu" and device2str(device="cuda:5") == "gpu:5"
    assert device2str(_Place(None)) == "cpu" and device2str(_Place(4)) == "gpu:4"
    sentinel = object()
    assert device2str(sentinel) is sentinel


def test_unknown_tensor_type_is_refused():
    from lowbit_quant_fa2_paddle_amd import _tensor
    with pytest.raises(TypeError):
        _tensor.ops_for(np.zeros(3))
