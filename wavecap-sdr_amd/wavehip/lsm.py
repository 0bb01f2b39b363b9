"""P25 Phase-1 CQPSK / LSM (linear simulcast) demodulator on the MI355X (row A12).

`CQPSKDemodulator` is the drop-in for wavecapsdr.decoders.p25.CQPSKDemodulator (p25.py:190-669: ctor
`(sample_rate=19200, symbol_rate=4800)`, `demodulate(iq) -> dibits uint8`); `LSMBank` runs n channels per
launch (one lane per channel -- the symbol clock, frequency loop and Gardner update feed back every
symbol).  One `demodulate` call = one reference call: the reference's AGC, NCO and 'same'-mode low-pass are
per call, so chunk boundaries are part of the result and calls are never split.  `LSMDemodulator` is an
alias (wavehip already has a Phase-2 class called CQPSKDemodulator in wavehip.cqpsk)."""

from __future__ import annotations

import ctypes as C
import logging

import numpy as np
from scipy import signal

from . import _lib

logger = logging.getLogger(__name__)

BASEBAND_CUTOFF_HZ = 7250
MMSE_NTAPS = 32
MMSE_NSTEPS = 128


def design_baseband_filter(sample_rate: float, num_taps: int = 63) -> np.ndarray:
    """p25.py:370-383."""
    wc = min(0.99, max(0.01, BASEBAND_CUTOFF_HZ / (sample_rate / 2)))
    return np.asarray(signal.firwin(num_taps, wc, window="hamming"), dtype=np.float32)


def generate_mmse_taps() -> np.ndarray:
    """p25.py:289-323: (NSTEPS+1) x 8 Hann-windowed sinc, float64 design stored as float32, each row then
    divided by its float32 sum."""
    mu = np.arange(MMSE_NSTEPS + 1) / MMSE_NSTEPS
    t = (np.arange(8) - 3)[None, :] - mu[:, None]
    on = np.abs(t) < 1e-6
    ts = np.where(on, 1.0, t)
    val = (np.sin(np.pi * ts) / (np.pi * ts)) * np.where(np.abs(ts) < 4, 0.5 * (1 + np.cos(np.pi * ts / 4)), 0.0)
    taps = np.where(on, 1.0, val).astype(np.float32)
    for row in taps:
        row /= np.sum(row)
    return taps


class LSMBank:
    def __init__(self, n_channels: int, sample_rate: int = 19200, symbol_rate: int = 4800,
                 max_samples_per_call: int = 1 << 17):
        self._torch = _lib.require_gpu()
        self.n_channels = int(n_channels)
        self.sample_rate, self.symbol_rate = sample_rate, symbol_rate
        self.samples_per_symbol = sample_rate / symbol_rate
        self.max_samples_per_call = int(max_samples_per_call)
        self._baseband_taps = np.ascontiguousarray(design_baseband_filter(sample_rate))
        self._mmse_taps = np.ascontiguousarray(generate_mmse_taps())
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_lsm_bank_destroy
        _lib.check(_lib.lib.wh_lsm_bank_create(C.byref(self._h), self.n_channels, float(self.samples_per_symbol),
                                               _lib.dptr(self._baseband_taps, "f32"), _lib.dptr(self._mmse_taps, "f32"),
                                               self.max_samples_per_call), "wh_lsm_bank_create")
        torch = self._torch
        self.cap = self.max_samples_per_call            # at most one symbol per sample
        self._dibits = torch.empty((self.n_channels, self.cap), dtype=torch.uint8, device="cuda")
        self._phases = None
        self._counts = torch.zeros(self.n_channels, dtype=torch.int32, device="cuda")

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self) -> None:
        _lib.check(_lib.lib.wh_lsm_bank_reset(self._h, _lib.stream_ptr(self._torch)), "wh_lsm_bank_reset")

    def reserve(self, max_samples_per_call: int) -> None:
        """Grow the bank for calls of up to `max_samples_per_call` samples per channel (allocates; all demodulator state
        is kept).  demodulate_device() does it by itself when a longer call arrives."""
        if max_samples_per_call <= self.max_samples_per_call:
            return
        torch = self._torch
        _lib.check(_lib.lib.wh_lsm_bank_reserve(self._h, int(max_samples_per_call), _lib.stream_ptr(torch)), "wh_lsm_bank_reserve")
        self.max_samples_per_call = self.cap = int(max_samples_per_call)
        self._dibits = torch.empty((self.n_channels, self.cap), dtype=torch.uint8, device="cuda")
        self._phases = None

    def state(self, channel: int = 0) -> dict:
        out = (C.c_double * 7)()
        _lib.check(_lib.lib.wh_lsm_bank_get_state(self._h, int(channel), out, _lib.stream_ptr(self._torch)),
                   "wh_lsm_bank_get_state")
        return dict(agc_gain=out[0], freq_offset=out[1], phase_acc=out[2], symbol_clock=out[3],
                    prev_symbol=complex(out[4], out[5]), clock_is_f32=bool(out[6]))

    def demodulate_device(self, iq_dev, want_phases: bool = False):
        """iq_dev complex64 GPU tensor [C, n] -> (dibits uint8 [C, cap], phases float32 [C, cap] | None,
        counts int32 [C]); the outputs are the bank's own buffers (valid until the next call)."""
        torch = self._torch
        assert iq_dev.is_cuda and iq_dev.dtype == torch.complex64 and iq_dev.dim() == 2
        assert iq_dev.shape[0] == self.n_channels and iq_dev.stride(1) == 1
        n = iq_dev.shape[1]
        if n > self.max_samples_per_call:
            # the reference takes a call of any length (decoders/p25.py:413) and is NOT invariant to where a stream is cut
            # (np.convolve(..., 'same') per call), so a long call is never split: the work buffers grow, state is kept
            self.reserve(max(n, self.max_samples_per_call + self.max_samples_per_call // 2))
        stride = iq_dev.stride(0) if self.n_channels > 1 else n
        if want_phases and self._phases is None:
            self._phases = torch.empty((self.n_channels, self.cap), dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib.wh_lsm_bank_run(self._h, iq_dev.data_ptr(), n, stride, self._dibits.data_ptr(),
                                            self._phases.data_ptr() if want_phases else None, self.cap,
                                            self._counts.data_ptr(), _lib.stream_ptr(torch)), "wh_lsm_bank_run")
        return self._dibits, (self._phases if want_phases else None), self._counts

    def demodulate(self, iq, want_phases: bool = False):
        """iq complex64 [C, n] host array = one reference call per channel -> list of dibit arrays (and
        phases)."""
        torch = self._torch
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        if x.shape[1] == 0:
            e = [np.array([], dtype=np.uint8) for _ in range(self.n_channels)]
            return (e, [np.array([], dtype=np.float32) for _ in e]) if want_phases else e
        d, ph, cnt = self.demodulate_device(torch.from_numpy(x).cuda(), want_phases)
        cnt = cnt.cpu().numpy()
        m = int(cnt.max())
        d = d[:, :m].cpu().numpy()
        dib = [d[c, :cnt[c]].copy() for c in range(self.n_channels)]
        if not want_phases:
            return dib
        ph = ph[:, :m].cpu().numpy()
        return dib, [ph[c, :cnt[c]].copy() for c in range(self.n_channels)]


class CQPSKDemodulator:
    """Single-channel drop-in for decoders/p25.py:190 (same ctor, attributes and demodulate contract)."""

    BASEBAND_CUTOFF_HZ = BASEBAND_CUTOFF_HZ
    MMSE_NTAPS = MMSE_NTAPS
    MMSE_NSTEPS = MMSE_NSTEPS

    def __init__(self, sample_rate: int = 19200, symbol_rate: int = 4800) -> None:
        self.sample_rate = sample_rate
        self.symbol_rate = symbol_rate
        self.samples_per_symbol = sample_rate / symbol_rate
        self._bank = LSMBank(1, sample_rate, symbol_rate)
        self._baseband_taps = self._bank._baseband_taps
        self._mmse_taps = self._bank._mmse_taps
        self.last_phases = np.zeros(0, dtype=np.float32)

    def reset(self) -> None:
        self._bank.reset()

    def state(self) -> dict:
        return self._bank.state(0)

    def demodulate(self, iq) -> np.ndarray:
        iq = np.asarray(iq)
        if iq.size == 0:
            return np.array([], dtype=np.uint8)
        if not np.iscomplexobj(iq):                       # p25.py:427-432
            logger.warning(f"CQPSK demodulate: expected complex IQ, got {iq.dtype}")
            if len(iq) % 2 == 0:
                iq = iq[::2] + 1j * iq[1::2]
            else:
                return np.array([], dtype=np.uint8)
        dib, ph = self._bank.demodulate(iq.astype(np.complex64, copy=False)[None, :], want_phases=True)
        self.last_phases = ph[0]
        return dib[0]


LSMDemodulator = CQPSKDemodulator
