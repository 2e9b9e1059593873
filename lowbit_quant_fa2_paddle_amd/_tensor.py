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

    def cumsum0_pad(self, t):
        """[0, cumsum(t)] as int32 (quant_per_block_varlen.py:95-100)."""
        return self.torch.nn.functional.pad(self.torch.cumsum(t, dim=0), (1, 0), value=0).to(self.torch.int32)


class PaddleOps:  # pragma: no cover - Paddle is absent from the build image; same primitives, untested here
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
        return self.paddle.empty(list(shape), dtype=dtype).to(like.place)

    def pad_last(self, t, n):
        return self.paddle.nn.functional.pad(t, [0, n], data_format="NCHW") if t.ndim != 4 else \
            self.paddle.concat([t, self.paddle.zeros(list(t.shape[:-1]) + [n], dtype=t.dtype)], axis=-1)

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

    def cumsum0_pad(self, t):
        return self.paddle.concat([self.paddle.zeros([1], dtype=t.dtype), self.paddle.cumsum(t, axis=0)]).astype(self.paddle.int32)


def device2str(type=None, index=None, *, device=None):
    """PaConvert helper the reference calls before every launch (paddle_utils.py:20-36): normalises a
    device spec to Paddle's 'gpu:N' / 'cpu' strings.  Same accepted inputs, duck-typed on the place."""
    type = device if device else type
    if isinstance(type, int):
        return f"gpu:{type}"
    if isinstance(type, str):
        if "cuda" in type:
            type = type.replace("cuda", "gpu")
        if "cpu" in type:
            return "cpu"
        if index is not None:
            type = f"{type}:{index}"
        return type
    if type is None:
        return "cpu"
    if hasattr(type, "is_cpu_place") and type.is_cpu_place():
        return "cpu"
    if hasattr(type, "get_device_id"):
        return f"gpu:{type.get_device_id()}"
    return type


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
