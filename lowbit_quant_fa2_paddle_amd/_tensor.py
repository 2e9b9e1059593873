"""Duck-typed tensor adaptor: the operator is host-framework agnostic.

The north star keeps host code on PaddlePaddle-ROCm; Paddle is not installed in the build image, so
torch-ROCm tensors are the tested carrier and Paddle tensors are handled through the same five
primitives (pointer, shape, strides, dtype, allocation).  Frameworks are plumbing only: device
memory, the current stream, zero padding of the head dimension - no arithmetic of the hot path.
"""
from __future__ import annotations

from . import _lib


def _is_torch(t):
    return type(t).__module__.startswith("torch")


def _is_paddle(t):
    return type(t).__module__.startswith("paddle")


class TorchOps:
    name = "torch"

    def __init__(self):
        import torch
        self.torch = torch
        self.int8, self.float32, self.uint8 = torch.int8, torch.float32, torch.uint8
        self.float16, self.bfloat16 = torch.float16, torch.bfloat16

    def dtype_code(self, t):
        if t.dtype == self.torch.float16:
            return _lib.LBFA_F16
        if t.dtype == self.torch.bfloat16:
            return _lib.LBFA_BF16
        return None

    def ptr(self, t):
        return t.data_ptr()

    def shape(self, t):
        return tuple(t.shape)

    def strides(self, t):
        return tuple(t.stride())

    def same_device(self, *ts):
        return all(t.device == ts[0].device for t in ts)

    def is_gpu(self, t):
        return t.is_cuda

    def empty(self, shape, dtype, like):
        return self.torch.empty(shape, dtype=dtype, device=like.device)

    def pad_last(self, t, n):
        return self.torch.nn.functional.pad(t, (0, n))

    def stream(self, t):
        return self.torch.cuda.current_stream(t.device).cuda_stream

    def device_guard(self, t):
        return self.torch.cuda.device(t.device)

    def cat0(self, ts):
        return self.torch.cat(ts, dim=0)

    def as_int32(self, t):
        """cu_seqlens may be int32 or int64 (src/core.py:418); the C ABI takes contiguous int32."""
        return t.to(self.torch.int32).contiguous()

    def tolist(self, t):
        """small device tensor -> Python list (one device-to-host copy; synchronises the stream)"""
        return t.tolist()

    def cumsum0_pad(self, t):
        """[0, cumsum(t)] as int32 (quant_per_block_varlen.py:95-100)."""
        return self.torch.nn.functional.pad(self.torch.cumsum(t, dim=0), (1, 0), value=0).to(self.torch.int32)


class PaddleOps:
    """Same primitives over Paddle tensors.  Paddle is absent from the build image: exercised by
    tests/test_paddle_adaptor_cpu.py against a stand-in module object, never against a live Paddle."""
    name = "paddle"

    def __init__(self):
        import paddle
        self.paddle = paddle
        self.int8, self.float32, self.uint8 = paddle.int8, paddle.float32, paddle.uint8
        self.float16, self.bfloat16 = paddle.float16, paddle.bfloat16

    def dtype_code(self, t):
        if t.dtype == self.paddle.float16:
            return _lib.LBFA_F16
        if t.dtype == self.paddle.bfloat16:
            return _lib.LBFA_BF16
        return None

    def ptr(self, t):
        return t.data_ptr()

    def shape(self, t):
        return tuple(t.shape)

    def strides(self, t):
        return tuple(t.strides)

    def same_device(self, *ts):
        return all(str(t.place) == str(ts[0].place) for t in ts)

    def is_gpu(self, t):
        return t.place.is_gpu_place()

    def empty(self, shape, dtype, like):
        # allocated ON `like`'s device: select it first - what the reference does in front of its own `paddle.empty` calls
        # (`paddle.device.set_device(device2str(v.place))`, src/core.py:276) - instead of allocating on the current device and
        # copying over with `.to(place)` (one extra allocation + copy per output)
        self.paddle.device.set_device(device2str(like.place))
        return self.paddle.empty(list(shape), dtype=dtype)

    def pad_last(self, t, n):
        # zero columns appended to the last axis, any rank (dense [B,H,S,D] / [B,S,H,D] and packed [T,H,D] tensors alike)
        return self.paddle.concat([t, self.paddle.zeros(list(t.shape[:-1]) + [n], dtype=t.dtype)], axis=-1)

    def stream(self, t):
        return self.paddle.device.current_stream().stream_base.raw_stream

    def device_guard(self, t):
        import contextlib
        # mirrors `paddle.device.set_device(device2str(v.place))` (src/core.py:276, paddle_utils.py:20-36)
        self.paddle.device.set_device(device2str(t.place))
        return contextlib.nullcontext()

    def cat0(self, ts):
        return self.paddle.concat(ts, axis=0)

    def as_int32(self, t):
        return t.astype(self.paddle.int32).contiguous()

    def tolist(self, t):
        return t.numpy().tolist()

    def cumsum0_pad(self, t):
        return self.paddle.concat([self.paddle.zeros([1], dtype=t.dtype), self.paddle.cumsum(t, axis=0)]).astype(self.paddle.int32)


def device2str(type=None, index=None, *, device=None):
    """Device spec -> Paddle's 'gpu:N' / 'cpu' string, the call the reference makes before every launch
    (`paddle.device.set_device(device2str(v.place))`, src/core.py:276; helper of paddle_utils.py).  Accepts what that helper
    accepts: a device index, a 'cuda[:N]' / 'gpu[:N]' / 'cpu' string (optionally with a separate index), None, or a place
    object (duck-typed: `is_cpu_place()` / `get_device_id()`); anything else is handed back unchanged."""
    spec = device if device else type
    if spec is None:
        return "cpu"
    if isinstance(spec, int):
        return f"gpu:{spec}"
    if isinstance(spec, str):
        if "cpu" in spec:
            return "cpu"
        name = spec.replace("cuda", "gpu")
        return name if index is None else f"{name}:{index}"
    is_cpu = getattr(spec, "is_cpu_place", None)
    if callable(is_cpu) and is_cpu():
        return "cpu"
    dev_id = getattr(spec, "get_device_id", None)
    return f"gpu:{dev_id()}" if callable(dev_id) else spec


_OPS = {}


def ops_for(t):
    if _is_torch(t):
        key = "torch"
    elif _is_paddle(t):
        key = "paddle"
    else:
        raise TypeError(f"unsupported tensor type {type(t)!r}: expected a paddle.Tensor or torch.Tensor")
    if key not in _OPS:
        _OPS[key] = TorchOps() if key == "torch" else PaddleOps()
    return _OPS[key]
