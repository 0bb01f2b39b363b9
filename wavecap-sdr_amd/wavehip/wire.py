"""int16 wire conventions (A1) on the GPU, host-array convenience wrappers.

pack_iq16  <- capture.py:102-116 (returns bytes like the reference)
pack_pcm16 <- capture.py:119-131
pack_f32   <- capture.py:134-144
unpack_iq16 <- cli.py:447-452 / harness.py:274
"""

from __future__ import annotations

import numpy as np

from . import _lib


def pack_iq16(samples: np.ndarray) -> bytes:
    if samples.size == 0:
        return b""
    torch = _lib.require_gpu()
    x = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.complex64)).cuda()
    out = torch.empty(2 * x.numel(), dtype=torch.int16, device=x.device)
    _lib.check(_lib.lib.wh_pack_cf32_i16(x.data_ptr(), out.data_ptr(), x.numel(), _lib.stream_ptr(torch)),
               "wh_pack_cf32_i16")
    return out.cpu().numpy().tobytes()


def pack_pcm16(samples: np.ndarray) -> bytes:
    if samples.size == 0:
        return b""
    torch = _lib.require_gpu()
    x = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float32)).cuda()
    out = torch.empty(x.numel(), dtype=torch.int16, device=x.device)
    _lib.check(_lib.lib.wh_pack_f32_pcm16(x.data_ptr(), out.data_ptr(), x.numel(), _lib.stream_ptr(torch)),
               "wh_pack_f32_pcm16")
    return out.cpu().numpy().tobytes()


def unpack_iq16(data) -> np.ndarray:
    i16 = np.frombuffer(data, dtype=np.int16) if isinstance(data, (bytes, bytearray, memoryview)) else \
        np.ascontiguousarray(data, dtype=np.int16)
    if i16.size == 0:
        return np.empty(0, dtype=np.complex64)
    torch = _lib.require_gpu()
    x = torch.from_numpy(i16.copy()).cuda()
    out = torch.empty(i16.size // 2, dtype=torch.complex64, device=x.device)
    _lib.check(_lib.lib.wh_unpack_i16_cf32(x.data_ptr(), out.data_ptr(), i16.size // 2, _lib.stream_ptr(torch)),
               "wh_unpack_i16_cf32")
    return out.cpu().numpy()


def pack_f32(samples: np.ndarray) -> bytes:
    if samples.size == 0:
        return b""
    torch = _lib.require_gpu()
    x = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float32)).cuda()
    out = torch.empty_like(x)
    _lib.check(_lib.lib.wh_clip_f32(x.data_ptr(), out.data_ptr(), x.numel(), _lib.stream_ptr(torch)), "wh_clip_f32")
    return out.cpu().numpy().tobytes()
