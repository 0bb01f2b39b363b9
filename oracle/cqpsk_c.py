"""ctypes wrapper of oracle/cqpsk_ref.c (row A12).  TEST INFRASTRUCTURE ONLY.

`CQPSKDemodulatorRef` mirrors dsp/p25/cqpsk.py:199-350 (ctor defaults, demodulate, reset);
`GardnerTEDRef` mirrors dsp/p25/symbol_timing.py:60-211 (process_block, reset).  Host-side designs
restate design_rrc_filter_phase2 (cqpsk.py:35-81) and calculate_loop_coefficients
(symbol_timing.py:34-57)."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
from scipy import signal

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcqpsk_ref.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "cqpsk_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        lib = C.CDLL(_SO)
        lib.cqpsk_ref_create.restype = C.c_void_p
        lib.cqpsk_ref_create.argtypes = [C.c_double, C.c_void_p, C.c_int, C.c_void_p] + [C.c_double] * 5
        lib.cqpsk_ref_demodulate.restype = C.c_int
        lib.cqpsk_ref_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.cqpsk_ref_reset.argtypes = [C.c_void_p]
        lib.cqpsk_ref_destroy.argtypes = [C.c_void_p]
        lib.gardner_ref_create.restype = C.c_void_p
        lib.gardner_ref_create.argtypes = [C.c_double] * 3
        lib.gardner_ref_process.restype = C.c_int
        lib.gardner_ref_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.gardner_ref_reset.argtypes = [C.c_void_p]
        lib.gardner_ref_destroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def loop_coefficients(loop_bw: float = 0.01, damping: float = 1.0) -> tuple[float, float]:
    """symbol_timing.py:34-57 / cqpsk.py:107-110."""
    theta = loop_bw / (damping + 1 / (4 * damping))
    d = 1 + 2 * damping * theta + theta ** 2
    return 4 * damping * theta / d, 4 * theta ** 2 / d


def design_rrc_filter_phase2(sps: float, num_taps: int = 65, alpha: float = 1.0) -> np.ndarray:
    """cqpsk.py:35-81 (unit-energy normalisation)."""
    if num_taps % 2 == 0:
        num_taps += 1
    t = (np.arange(num_taps) - (num_taps - 1) / 2) / sps
    h = np.zeros(num_taps, dtype=np.float64)
    for i, ti in enumerate(t):
        if ti == 0:
            h[i] = 1 - alpha + 4 * alpha / np.pi
        elif abs(ti * 4 * alpha) == 1:
            h[i] = (alpha / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha))
                                           + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            num = np.sin(np.pi * ti * (1 - alpha)) + 4 * alpha * ti * np.cos(np.pi * ti * (1 + alpha))
            den = np.pi * ti * (1 - (4 * alpha * ti) ** 2)
            h[i] = num / den if abs(den) > 1e-10 else 0
    h = h / np.sqrt(np.sum(h ** 2))
    return np.asarray(h, dtype=np.float32)


class CQPSKDemodulatorRef:
    def __init__(self, sample_rate: int = 48000, symbol_rate: int = 12000, rrc_alpha: float = 1.0,
                 rrc_taps: int = 65, carrier_loop_bw: float = 0.01, timing_loop_bw: float = 0.01):
        lib = _load()
        self.samples_per_symbol = sample_rate / symbol_rate
        self.rrc = design_rrc_filter_phase2(self.samples_per_symbol, rrc_taps, rrc_alpha)
        zi = np.ascontiguousarray(signal.lfilter_zi(self.rrc, 1.0), dtype=np.float64)
        ckp, cki = loop_coefficients(carrier_loop_bw, 0.707)
        tkp, tki = loop_coefficients(timing_loop_bw, 1.0)
        self._h = lib.cqpsk_ref_create(self.samples_per_symbol, self.rrc.ctypes.data, len(self.rrc), zi.ctypes.data,
                                       ckp, cki, 0.1, tkp, tki)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.cqpsk_ref_destroy(self._h)
            self._h = None

    def reset(self):
        _lib.cqpsk_ref_reset(self._h)

    def demodulate(self, iq, want_symbols: bool = False):
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        n = x.shape[0]
        if n == 0:
            return (np.array([], np.uint8), np.zeros(0, np.complex128)) if want_symbols else np.array([], np.uint8)
        cap = n + 8
        d = np.empty(cap, dtype=np.uint8)
        s = np.empty(cap, dtype=np.complex128)
        cnt = _lib.cqpsk_ref_demodulate(self._h, x.ctypes.data, n, d.ctypes.data, s.ctypes.data, cap)
        return (d[:cnt].copy(), s[:cnt].copy()) if want_symbols else d[:cnt].copy()


class GardnerTEDRef:
    def __init__(self, samples_per_symbol: float, loop_bw: float = 0.01, damping: float = 1.0):
        lib = _load()
        kp, ki = loop_coefficients(loop_bw, damping)
        self._h = lib.gardner_ref_create(float(samples_per_symbol), kp, ki)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.gardner_ref_destroy(self._h)
            self._h = None

    def reset(self):
        _lib.gardner_ref_reset(self._h)

    def process_block(self, samples):
        x = np.ascontiguousarray(samples, dtype=np.float32)
        n = x.shape[0]
        cap = n + 8
        s = np.empty(cap, dtype=np.float64)
        e = np.empty(cap, dtype=np.float64)
        cnt = _lib.gardner_ref_process(self._h, x.ctypes.data, n, s.ctypes.data, e.ctypes.data, cap)
        return s[:cnt].copy(), e[:cnt].copy()
