"""ctypes wrapper of oracle/lsm_ref.c (row A12, the LSM demodulator of decoders/p25.py:190-669).
TEST INFRASTRUCTURE ONLY.

`LSMDemodulatorRef(sample_rate, symbol_rate, flavour)` mirrors the reference class' ctor / demodulate;
the host-side designs restate _design_baseband_filter (:370-383) and _generate_mmse_taps (:289-323)."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
from scipy import signal

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblsm_ref.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "lsm_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        lib = C.CDLL(_SO)
        lib.lsm_ref_create.restype = C.c_void_p
        lib.lsm_ref_create.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        lib.lsm_ref_demodulate.restype = C.c_int
        lib.lsm_ref_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.lsm_ref_get_state.argtypes = [C.c_void_p, C.c_void_p]
        lib.lsm_ref_destroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def baseband_taps(sample_rate: float, cutoff_hz: float = 7250.0, num_taps: int = 63) -> np.ndarray:
    """p25.py:370-383: Hamming firwin at 7250 Hz, cutoff clamped to [0.01, 0.99] of Nyquist."""
    wc = min(0.99, max(0.01, cutoff_hz / (sample_rate / 2)))
    return np.asarray(signal.firwin(num_taps, wc, window="hamming"), dtype=np.float32)


def mmse_table() -> np.ndarray:
    """p25.py:289-323: 129 x 8 Hann-windowed sinc, rows normalised to unit sum in float32."""
    taps = np.zeros((129, 8), dtype=np.float32)
    for step in range(129):
        mu = step / 128
        for tap in range(8):
            t = tap - 3 - mu
            if abs(t) < 1e-6:
                taps[step, tap] = 1.0
            else:
                s = np.sin(np.pi * t) / (np.pi * t)
                w = 0.5 * (1 + np.cos(np.pi * t / 4)) if abs(t) < 4 else 0
                taps[step, tap] = s * w
        tot = np.sum(taps[step])
        if abs(tot) > 1e-6:
            taps[step] /= tot
    return taps


class LSMDemodulatorRef:
    def __init__(self, sample_rate: int = 19200, symbol_rate: int = 4800, flavour: int = 1):
        self._lib = _load()
        self.sample_rate, self.symbol_rate = sample_rate, symbol_rate
        self.samples_per_symbol = sample_rate / symbol_rate
        self.lpf = np.ascontiguousarray(baseband_taps(sample_rate))
        self.mmse = np.ascontiguousarray(mmse_table())
        self._args = (C.c_double(self.samples_per_symbol), self.lpf.ctypes.data, self.mmse.ctypes.data, flavour)
        self._h = self._lib.lsm_ref_create(*self._args)
        self.last_phases = np.zeros(0, dtype=np.float32)
        self.last_filtered = np.zeros(0, dtype=np.complex64)

    def reset(self):
        self._lib.lsm_ref_destroy(self._h)
        self._h = self._lib.lsm_ref_create(*self._args)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.lsm_ref_destroy(self._h)
            self._h = None

    def demodulate(self, iq: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        n = x.size
        if n == 0:
            return np.array([], dtype=np.uint8)
        cap = int(n / self.samples_per_symbol * 1.1) + 8
        dib = np.zeros(cap, dtype=np.uint8)
        ph = np.zeros(cap, dtype=np.float32)
        filt = np.zeros(n, dtype=np.complex64)
        cnt = self._lib.lsm_ref_demodulate(self._h, x.ctypes.data, n, dib.ctypes.data, ph.ctypes.data,
                                           filt.ctypes.data, cap)
        assert 0 <= cnt <= cap
        self.last_phases, self.last_filtered = ph[:cnt].copy(), filt
        return dib[:cnt].copy()

    def state(self) -> dict:
        s = np.zeros(7, dtype=np.float64)
        self._lib.lsm_ref_get_state(self._h, s.ctypes.data)
        return dict(agc_gain=s[0], freq_offset=s[1], phase_acc=s[2], symbol_clock=s[3],
                    prev_symbol=complex(s[4], s[5]), f32mode=bool(s[6]))
