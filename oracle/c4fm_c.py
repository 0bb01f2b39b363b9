"""ctypes wrapper of the C oracle for the P25 C4FM demodulator (oracle/c4fm_ref.c).

TEST INFRASTRUCTURE ONLY (see oracle/ref_np.py header).  `C4FMDemodulatorRef` mirrors
C4FMDemodulator (c4fm.py:2379-2528): ctor `(sample_rate=19200, symbol_rate=4800, wide_pulse=False)`,
`demodulate(iq) -> (dibits uint8, soft float32)`, `reset()`.  Filter designs restate
design_baseband_lpf (c4fm.py:95-132) and design_rrc_filter (c4fm.py:135-183)."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
from scipy import signal

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libc4fm_ref.so")
_TAPS = os.path.join(os.path.dirname(_HERE), "tests", "golden", "mmse_interp_taps_f32.npy")


def _load():
    src = os.path.join(_HERE, "c4fm_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    lib = C.CDLL(_SO)
    lib.c4fm_ref_create.restype = C.c_void_p
    lib.c4fm_ref_create.argtypes = [C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.c4fm_ref_demodulate.restype = C.c_int
    lib.c4fm_ref_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.c4fm_ref_reset.argtypes = [C.c_void_p]
    lib.c4fm_ref_destroy.argtypes = [C.c_void_p]
    lib.c4fm_ref_get_state.argtypes = [C.c_void_p, C.c_void_p]
    return lib


_lib = None


def design_baseband_lpf(sample_rate: float, passband_hz: float = 5200.0, stopband_hz: float = 6500.0,
                        num_taps: int = 63) -> np.ndarray:
    """c4fm.py:95-132.  The reference passes remez(..., Hz=sample_rate); scipy >= 1.14 removed
    that keyword, the call raises and the reference's `except Exception` falls back to a Hamming
    windowed sinc at the passband edge -- which is therefore the filter under the oracle
    semantics (scipy 1.15.3).  Same try/except here so the behaviour tracks the installed scipy."""
    try:
        h = signal.remez(num_taps, [0, passband_hz, stopband_hz, sample_rate / 2.0], [1, 0], Hz=sample_rate)
    except Exception:
        h = signal.firwin(num_taps, passband_hz, fs=sample_rate, window="hamming")
    return np.asarray(h, dtype=np.float32)


def design_rrc_filter(sps: float, num_taps: int, alpha: float = 0.2) -> np.ndarray:
    """c4fm.py:135-183 (normalised to unit DC gain)."""
    if num_taps % 2 == 0:
        num_taps += 1
    n = np.arange(num_taps) - (num_taps - 1) / 2
    t = n / sps
    h = np.zeros(num_taps, dtype=np.float64)
    for i, ti in enumerate(t):
        if ti == 0:
            h[i] = 1 - alpha + 4 * alpha / np.pi
        elif abs(ti) == 1 / (4 * alpha):
            h[i] = (alpha / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha))
                                           + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            num = np.sin(np.pi * ti * (1 - alpha)) + 4 * alpha * ti * np.cos(np.pi * ti * (1 + alpha))
            den = np.pi * ti * (1 - (4 * alpha * ti) ** 2)
            h[i] = num / den
    return (h / np.sum(h)).astype(np.float32)


class C4FMDemodulatorRef:
    def __init__(self, sample_rate: int = 19200, symbol_rate: int = 4800, wide_pulse: bool = False,
                 atan_mode: int = 1, **_kw):
        global _lib
        if _lib is None:
            _lib = _load()
        self.sample_rate, self.symbol_rate = sample_rate, symbol_rate
        self.samples_per_symbol = sample_rate / symbol_rate
        pb, sb, alpha = (10000.0, 12000.0, 0.5) if wide_pulse else (5200.0, 6500.0, 0.2)
        self.lpf = design_baseband_lpf(sample_rate, pb, sb)
        self.rrc = design_rrc_filter(self.samples_per_symbol, int(16 * self.samples_per_symbol) + 1, alpha)
        self.taps = np.ascontiguousarray(np.load(_TAPS), dtype=np.float32)
        assert self.taps.shape == (129, 8)
        self._h = _lib.c4fm_ref_create(self.samples_per_symbol, self.lpf.ctypes.data, len(self.lpf),
                                       self.rrc.ctypes.data, len(self.rrc), self.taps.ctypes.data, atan_mode)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.c4fm_ref_destroy(self._h)
            self._h = None

    def reset(self) -> None:
        _lib.c4fm_ref_reset(self._h)

    def demodulate(self, iq):
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        n = x.shape[0]
        if n == 0:
            return np.array([], dtype=np.uint8), np.array([], dtype=np.float32)
        cap = n // 4 + 16
        d = np.empty(cap, dtype=np.uint8)
        s = np.empty(cap, dtype=np.float32)
        cnt = _lib.c4fm_ref_demodulate(self._h, x.ctypes.data, n, d.ctypes.data, s.ctypes.data, cap)
        return d[:cnt].copy(), s[:cnt].copy()

    def state(self) -> dict:
        out = np.zeros(8, dtype=np.float64)
        _lib.c4fm_ref_get_state(self._h, out.ctypes.data)
        return dict(sync_count=int(out[0]), fine_sync=bool(out[1]), pll=out[2], gain=out[3], sample_point=out[4],
                    sp_np64=bool(out[5]), buffer_pointer=int(out[6]))
