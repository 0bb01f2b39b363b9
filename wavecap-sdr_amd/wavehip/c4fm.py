"""P25 Phase-1 C4FM demodulator on the MI355X: drop-in for
wavecapsdr.dsp.p25.c4fm.C4FMDemodulator (c4fm.py:2379-2807) plus a batched bank.

`C4FMDemodulator(sample_rate=19200, symbol_rate=4800, wide_pulse=False, **kwargs)`,
`demodulate(iq) -> (dibits uint8[k], soft float32[k])`, `reset()` -- state (filter history,
65536-sample phase buffer, symbol clock, equaliser, sync detectors) lives on the device
between calls, exactly one instance per channel like the reference (control_channel.py:230).
`C4FMBank` runs C independent demodulators per launch (BASELINE.json configs[3]).

Call size: one `demodulate(iq)` here is one `demodulate(iq)` there, whatever `len(iq)` is -- the reference's block
processing is not invariant to where a stream is cut (c4fm.py:704-728, 2621-2770; the control-channel monitor calls it
with 72 000-75 000 samples, trunking/system.py:1548-1549), so a call is never split; `max_samples_per_call` only sizes
the device workspaces and grows on demand.

Filter designs are the reference's own host-side designs (design_baseband_lpf c4fm.py:95-132,
design_rrc_filter :135-183); the 129x8 MMSE interpolator table (:907-2202, GNU Radio /
SDRTrunk constants) ships as a binary data file."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
from scipy import signal

from . import _lib

SYMBOL_RATE = 4800
_TAPS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mmse_interp_taps_f32.npy")


def design_baseband_lpf(sample_rate: float, passband_hz: float = 5200.0, stopband_hz: float = 6500.0,
                        num_taps: int = 63) -> np.ndarray:
    """c4fm.py:95-132, including its fallback: remez(..., Hz=) raises on scipy >= 1.14 and the
    reference then designs a Hamming windowed sinc -- same try/except so both stay in step."""
    try:
        h = signal.remez(num_taps, [0, passband_hz, stopband_hz, sample_rate / 2.0], [1, 0], Hz=sample_rate)
    except Exception:
        h = signal.firwin(num_taps, passband_hz, fs=sample_rate, window="hamming")
    return np.asarray(h, dtype=np.float32)


def design_rrc_filter(samples_per_symbol: float, num_taps: int = 101, alpha: float = 0.2) -> np.ndarray:
    """c4fm.py:135-183 (DC gain normalised to 1)."""
    if num_taps % 2 == 0:
        num_taps += 1
    t = (np.arange(num_taps) - (num_taps - 1) / 2) / samples_per_symbol
    h = np.zeros(num_taps, dtype=np.float64)
    for i, ti in enumerate(t):
        if ti == 0:
            h[i] = 1 - alpha + 4 * alpha / np.pi
        elif abs(ti) == 1 / (4 * alpha):
            h[i] = (alpha / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha))
                                           + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
        else:
            num = np.sin(np.pi * ti * (1 - alpha)) + 4 * alpha * ti * np.cos(np.pi * ti * (1 + alpha))
            h[i] = num / (np.pi * ti * (1 - (4 * alpha * ti) ** 2))
    return (h / np.sum(h)).astype(np.float32)


class C4FMBank:
    """C independent C4FM demodulators, one wavefront each for the feedback part."""

    def __init__(self, n_channels: int, sample_rate: int = 19200, symbol_rate: int = SYMBOL_RATE,
                 wide_pulse: bool = False, max_samples_per_call: int = 16384):
        self._torch = _lib.require_gpu()
        self.n_channels = int(n_channels)
        self.sample_rate, self.symbol_rate = sample_rate, symbol_rate
        self.samples_per_symbol = sample_rate / symbol_rate
        self.wide_pulse = wide_pulse
        self.max_samples_per_call = int(max_samples_per_call)
        pb, sb, alpha = (10000.0, 12000.0, 0.5) if wide_pulse else (5200.0, 6500.0, 0.2)
        self._baseband_lpf = design_baseband_lpf(sample_rate, passband_hz=pb, stopband_hz=sb)
        self._rrc_filter = design_rrc_filter(self.samples_per_symbol,
                                             num_taps=int(16 * self.samples_per_symbol) + 1, alpha=alpha)
        taps = np.ascontiguousarray(np.load(_TAPS_FILE), dtype=np.float32)
        assert taps.shape == (129, 8)
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_c4fm_bank_destroy
        _lib.check(_lib.lib.wh_c4fm_bank_create(
            C.byref(self._h), self.n_channels, float(self.samples_per_symbol),
            _lib.dptr(self._baseband_lpf, "f32"), len(self._baseband_lpf),
            _lib.dptr(self._rrc_filter, "f32"), len(self._rrc_filter), _lib.dptr(taps, "f32"),
            self.max_samples_per_call), "wh_c4fm_bank_create")
        self._counts = self._torch.zeros(self.n_channels, dtype=self._torch.int32, device="cuda")
        self._size_outputs()

    def _size_outputs(self) -> None:
        torch = self._torch
        self.out_cap = self.max_samples_per_call // 4 + 16
        self._dibits = torch.empty((self.n_channels, self.out_cap), dtype=torch.uint8, device="cuda")
        self._soft = torch.empty((self.n_channels, self.out_cap), dtype=torch.float32, device="cuda")

    def reserve(self, max_samples_per_call: int) -> None:
        """Grow the bank for calls of up to `max_samples_per_call` samples per channel (allocates; all demodulator state
        is kept).  Done implicitly by a longer call; call it up front to keep allocation out of the first long call."""
        if max_samples_per_call <= self.max_samples_per_call:
            return
        _lib.check(_lib.lib.wh_c4fm_bank_reserve(self._h, int(max_samples_per_call), _lib.stream_ptr(self._torch)),
                   "wh_c4fm_bank_reserve")
        self.max_samples_per_call = int(max_samples_per_call)
        self._size_outputs()

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self) -> None:
        _lib.check(_lib.lib.wh_c4fm_bank_reset(self._h, _lib.stream_ptr(self._torch)), "wh_c4fm_bank_reset")

    def demodulate_device(self, iq_dev):
        """iq_dev: complex64 GPU tensor [C, n] = ONE reference call per channel (a longer n than any before grows the
        workspaces first).  Returns GPU tensors
        (dibits uint8 [C, cap], soft float32 [C, cap], counts int32 [C]); rows are valid up to counts[c]
        and are overwritten by the next call."""
        torch = self._torch
        assert iq_dev.is_cuda and iq_dev.dtype == torch.complex64 and iq_dev.dim() == 2
        assert iq_dev.shape[0] == self.n_channels and iq_dev.stride(1) == 1
        n = iq_dev.shape[1]
        if n > self.max_samples_per_call:
            self.reserve(n)
        row_stride = iq_dev.stride(0) if self.n_channels > 1 else n   # a size-1 dim may report any stride
        _lib.check(_lib.lib.wh_c4fm_bank_run(self._h, iq_dev.data_ptr(), n, row_stride,
                                             self._dibits.data_ptr(), self._soft.data_ptr(), self.out_cap,
                                             self._counts.data_ptr(), _lib.stream_ptr(torch)), "wh_c4fm_bank_run")
        return self._dibits, self._soft, self._counts

    def demodulate(self, iq) -> list[tuple[np.ndarray, np.ndarray]]:
        """iq: complex array [C, n] on the host -> [(dibits, soft)] per channel."""
        torch = self._torch
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        if x.shape[1] == 0:
            return [(np.array([], dtype=np.uint8), np.array([], dtype=np.float32))] * self.n_channels
        d, sf, cnt = self.demodulate_device(torch.from_numpy(x).cuda())
        d, sf, cnt = d.cpu().numpy(), sf.cpu().numpy(), cnt.cpu().numpy()
        return [(d[c, :cnt[c]].copy(), sf[c, :cnt[c]].copy()) for c in range(self.n_channels)]


class C4FMDemodulator:
    """Single-channel drop-in (c4fm.py:2412-2528)."""

    def __init__(self, sample_rate: int = 19200, symbol_rate: int = SYMBOL_RATE, wide_pulse: bool = False,
                 **kwargs):
        self.sample_rate, self.symbol_rate, self.wide_pulse = sample_rate, symbol_rate, wide_pulse
        self.samples_per_symbol = sample_rate / symbol_rate
        self._bank = C4FMBank(1, sample_rate, symbol_rate, wide_pulse,
                              max_samples_per_call=int(kwargs.get("max_samples_per_call", 81920)))

    @property
    def _ted_phase(self) -> float:  # c4fm.py:2523-2526
        return 0.0

    def reset(self) -> None:
        self._bank.reset()

    def demodulate(self, iq):
        if len(iq) == 0:  # c4fm.py:2546-2550
            return np.array([], dtype=np.uint8), np.array([], dtype=np.float32)
        return self._bank.demodulate(np.asarray(iq)[None, :])[0]


def c4fm_demod_simple(iq, sample_rate: int = 19200, symbol_rate: int = SYMBOL_RATE) -> np.ndarray:
    """c4fm.py:2995-3015."""
    return C4FMDemodulator(sample_rate=sample_rate, symbol_rate=symbol_rate).demodulate(iq)[0]
