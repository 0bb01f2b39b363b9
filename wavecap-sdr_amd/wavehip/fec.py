"""BCH(63,16,23) decoder for the P25 NID on the MI355X: drop-in for
wavecapsdr.dsp.fec.bch.bch_decode / BCH_63_16_23.decode (bch.py:533-658).

The reference runs syndromes -> Berlekamp-Massey -> Chien search -> re-check per word.  The code has only 2^16
codewords and minimum distance 23, so bounded-distance decoding (t = 11) is the same function as "the unique
codeword within 11 bit errors, if any": the device compares a word with all 65 536 codewords (xor + popcount, one
workgroup per word, the 512 KiB table stays in L2) -- identical (data, error-count) results, batched."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

N, K, T = 63, 16, 11
GENERATOR_POLY = 0o6331141367235453        # g(x) of BCH(63,16): lcm of the minimal polynomials of a^1..a^22, GF(2^6) mod x^6+x+1
MESSAGE_NOT_CORRECTED = -1


def codeword_table() -> np.ndarray:
    """uint64[65536]: systematic codeword of every 16-bit word d: d(x) x^47 + (d(x) x^47 mod g(x)); bit 62 is the
    first transmitted bit (the reference's codeword[0])."""
    d = np.arange(1 << K, dtype=np.uint64)
    rem = d << np.uint64(N - K)
    g = np.uint64(GENERATOR_POLY)
    for bit in range(N - 1, N - K - 1, -1):               # cancel x^62 .. x^47
        hit = (rem >> np.uint64(bit)) & np.uint64(1)
        rem ^= (g << np.uint64(bit - (N - K))) * hit
    return (d << np.uint64(N - K)) | rem


def pack_bits(bits) -> np.ndarray:
    """uint8 bit arrays [..., 63] (first bit = codeword[0]) -> uint64 words."""
    b = np.asarray(bits, dtype=np.uint64)[..., :N]
    w = np.zeros(b.shape[:-1], dtype=np.uint64)
    for i in range(N):
        w = (w << np.uint64(1)) | (b[..., i] & np.uint64(1))
    return w


class BCHDecoder:
    def __init__(self):
        self._torch = _lib.require_gpu()
        self._table = np.ascontiguousarray(codeword_table())
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_bch_destroy
        _lib.check(_lib.lib.wh_bch_create(C.byref(self._h), self._table.ctypes.data), "wh_bch_create")

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def decode_device(self, words_dev, tracked_dev=None):
        """words_dev int64/uint64 GPU tensor [n] (63-bit words) -> (data int32 [n], errors int32 [n]) on the GPU."""
        torch = self._torch
        assert words_dev.is_cuda and words_dev.element_size() == 8 and words_dev.is_contiguous()
        n = words_dev.numel()
        data = torch.empty(n, dtype=torch.int32, device="cuda")
        err = torch.empty(n, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib.wh_bch_decode(self._h, words_dev.data_ptr(), n,
                                          None if tracked_dev is None else tracked_dev.data_ptr(), data.data_ptr(),
                                          err.data_ptr(), _lib.stream_ptr(torch)), "wh_bch_decode")
        return data, err

    def decode_batch(self, words, tracked_nac=None):
        """words: uint64[n] packed words or uint8[n, 63] bit arrays; tracked_nac: None, an int or int[n]."""
        torch = self._torch
        w = np.asarray(words)
        if w.dtype != np.uint64:
            w = pack_bits(w)
        w = np.ascontiguousarray(np.atleast_1d(w)).view(np.int64)
        tr = None
        if tracked_nac is not None:
            tr = torch.from_numpy(np.broadcast_to(np.asarray(tracked_nac, dtype=np.int32), w.shape).copy()).cuda()
        d, e = self.decode_device(torch.from_numpy(w).cuda(), tr)
        return d.cpu().numpy(), e.cpu().numpy()

    def decode(self, codeword, tracked_nac=None) -> tuple[int, int]:
        if len(codeword) < N:                                # bch.py:584-586
            return 0, MESSAGE_NOT_CORRECTED
        d, e = self.decode_batch(np.asarray(codeword, dtype=np.uint8)[None, :N], tracked_nac if tracked_nac else None)
        return int(d[0]), int(e[0])


_decoder: BCHDecoder | None = None


def bch_decode(codeword, tracked_nac=None) -> tuple[int, int]:
    """bch.py:644-658."""
    global _decoder
    if _decoder is None:
        _decoder = BCHDecoder()
    return _decoder.decode(codeword, tracked_nac)
