"""P25 Phase-2 CQPSK demodulator and Gardner TED on the MI355X (row A12).

Drop-ins for wavecapsdr.dsp.p25.cqpsk.CQPSKDemodulator (cqpsk.py:199-350: same ctor defaults,
`demodulate(iq) -> dibits uint8`, `reset()`, `get_carrier_offset()` is not tracked on the host)
and wavecapsdr.dsp.p25.symbol_timing.GardnerTED (symbol_timing.py:60-211: `process_block(samples)
-> (symbols float64, errors float64)`, `reset()`), plus batched banks (one lane per channel --
both algorithms are per-sample feedback loops).  Float64 throughout, like the reference."""

from __future__ import annotations

import ctypes as C

import numpy as np
from scipy import signal

from . import _lib


def calculate_loop_coefficients(samples_per_symbol: float, loop_bw: float = 0.01, damping: float = 1.0):
    """symbol_timing.py:34-57 (samples_per_symbol is unused by the reference formula as well)."""
    theta = loop_bw / (damping + 1 / (4 * damping))
    d = 1 + 2 * damping * theta + theta ** 2
    return 4 * damping * theta / d, 4 * theta ** 2 / d


def design_rrc_filter_phase2(samples_per_symbol: float, num_taps: int = 65, alpha: float = 1.0) -> np.ndarray:
    """cqpsk.py:35-81."""
    if num_taps % 2 == 0:
        num_taps += 1
    t = (np.arange(num_taps) - (num_taps - 1) / 2) / samples_per_symbol
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
    return np.asarray(h / np.sqrt(np.sum(h ** 2)), dtype=np.float32)


class CQPSKBank:
    def __init__(self, n_channels: int, sample_rate: int = 48000, symbol_rate: int = 12000, rrc_alpha: float = 1.0,
                 rrc_taps: int = 65, carrier_loop_bw: float = 0.01, timing_loop_bw: float = 0.01,
                 max_samples_per_call: int = 1 << 16):
        self._torch = _lib.require_gpu()
        self.n_channels = int(n_channels)
        self.sample_rate, self.symbol_rate = sample_rate, symbol_rate
        self.samples_per_symbol = sample_rate / symbol_rate
        self.max_samples_per_call = int(max_samples_per_call)
        self._rrc = design_rrc_filter_phase2(self.samples_per_symbol, rrc_taps, rrc_alpha)
        zi = np.ascontiguousarray(signal.lfilter_zi(self._rrc, 1.0), dtype=np.float64)
        ckp, cki = calculate_loop_coefficients(0.0, carrier_loop_bw, 0.707)      # cqpsk.py:107-110
        tkp, tki = calculate_loop_coefficients(self.samples_per_symbol, timing_loop_bw, 1.0)
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_cqpsk_bank_destroy
        _lib.check(_lib.lib.wh_cqpsk_bank_create(C.byref(self._h), self.n_channels, float(self.samples_per_symbol),
                                                 _lib.dptr(self._rrc, "f32"), len(self._rrc), _lib.dptr(zi, "f64"),
                                                 ckp, cki, 0.1, tkp, tki, self.max_samples_per_call),
                   "wh_cqpsk_bank_create")
        self._counts = self._torch.zeros(self.n_channels, dtype=self._torch.int32, device="cuda")
        self._symbols = None
        self._size_outputs()

    def _size_outputs(self) -> None:
        torch = self._torch
        self.cap = int(self.max_samples_per_call / (self.samples_per_symbol * 0.5)) + 8
        self._dibits = torch.empty((self.n_channels, self.cap), dtype=torch.uint8, device="cuda")
        self._symbols = None        # complex128 [C, cap], allocated when symbols are first asked for

    def reserve(self, max_samples_per_call: int) -> None:
        """Grow the bank for calls of up to `max_samples_per_call` samples per channel (allocates; state is kept)."""
        if max_samples_per_call <= self.max_samples_per_call:
            return
        _lib.check(_lib.lib.wh_cqpsk_bank_reserve(self._h, int(max_samples_per_call), _lib.stream_ptr(self._torch)),
                   "wh_cqpsk_bank_reserve")
        self.max_samples_per_call = int(max_samples_per_call)
        self._size_outputs()

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self) -> None:
        _lib.check(_lib.lib.wh_cqpsk_bank_reset(self._h, _lib.stream_ptr(self._torch)), "wh_cqpsk_bank_reset")

    def demodulate_device(self, iq_dev, want_symbols: bool = False):
        torch = self._torch
        assert iq_dev.is_cuda and iq_dev.dtype == torch.complex64 and iq_dev.dim() == 2
        assert iq_dev.shape[0] == self.n_channels and iq_dev.stride(1) == 1
        n = iq_dev.shape[1]
        if n > self.max_samples_per_call:       # one call of the reference == one launch, whatever its length
            self.reserve(n)
        if want_symbols and self._symbols is None:
            self._symbols = torch.empty((self.n_channels, self.cap), dtype=torch.complex128, device="cuda")
        stride = iq_dev.stride(0) if self.n_channels > 1 else n
        _lib.check(_lib.lib.wh_cqpsk_bank_run(self._h, iq_dev.data_ptr(), n, stride, self._dibits.data_ptr(),
                                              self._symbols.data_ptr() if want_symbols else None, self.cap,
                                              self._counts.data_ptr(), _lib.stream_ptr(torch)), "wh_cqpsk_bank_run")
        return self._dibits, self._symbols, self._counts

    def demodulate(self, iq, want_symbols: bool = False):
        """iq: complex [C, n] on the host -> per channel dibits (and complex128 symbols).  The whole block is ONE
        reference call: the chain re-seeds its matched filter with zi * iq[0] per call (cqpsk.py:283-285), so a block is
        never cut into pieces here; a block longer than `max_samples_per_call` grows the bank."""
        torch = self._torch
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        if x.shape[1] == 0:
            e = np.array([], np.uint8)
            return [(e, np.zeros(0, np.complex128)) if want_symbols else e for _ in range(self.n_channels)]
        d, sy, cnt = self.demodulate_device(torch.from_numpy(x).cuda(), want_symbols)
        d, cnt = d.cpu().numpy(), cnt.cpu().numpy()
        sy = sy.cpu().numpy() if want_symbols else None
        if want_symbols:
            return [(d[c, :cnt[c]].copy(), sy[c, :cnt[c]].copy()) for c in range(self.n_channels)]
        return [d[c, :cnt[c]].copy() for c in range(self.n_channels)]


class CQPSKDemodulator:
    """Single-channel drop-in (cqpsk.py:224-306)."""

    def __init__(self, sample_rate: int = 48000, symbol_rate: int = 12000, rrc_alpha: float = 1.0,
                 rrc_taps: int = 65, carrier_loop_bw: float = 0.01, timing_loop_bw: float = 0.01):
        self.sample_rate, self.symbol_rate = sample_rate, symbol_rate
        self.samples_per_symbol = sample_rate / symbol_rate
        self._bank = CQPSKBank(1, sample_rate, symbol_rate, rrc_alpha, rrc_taps, carrier_loop_bw, timing_loop_bw)

    def reset(self) -> None:
        self._bank.reset()

    def demodulate(self, iq):
        if len(iq) == 0:
            return np.array([], dtype=np.uint8)
        return self._bank.demodulate(np.asarray(iq)[None, :])[0]


class GardnerBank:
    def __init__(self, n_channels: int, samples_per_symbol: float, loop_bw: float = 0.01, damping: float = 1.0):
        self._torch = _lib.require_gpu()
        self.n_channels = int(n_channels)
        self.samples_per_symbol = float(samples_per_symbol)
        kp, ki = calculate_loop_coefficients(samples_per_symbol, loop_bw, damping)
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_gardner_bank_destroy
        _lib.check(_lib.lib.wh_gardner_bank_create(C.byref(self._h), self.n_channels, self.samples_per_symbol, kp, ki),
                   "wh_gardner_bank_create")

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self) -> None:
        _lib.check(_lib.lib.wh_gardner_bank_reset(self._h, _lib.stream_ptr(self._torch)), "wh_gardner_bank_reset")

    def process_device(self, x_dev):
        """x_dev: float32 GPU tensor [C, n] -> (symbols f64 [C, cap], errors f64 [C, cap], counts int32 [C])."""
        torch = self._torch
        assert x_dev.is_cuda and x_dev.dtype == torch.float32 and x_dev.dim() == 2 and x_dev.stride(1) == 1
        n = x_dev.shape[1]
        cap = int(n / (self.samples_per_symbol * 0.5)) + 8
        sym = torch.empty((self.n_channels, cap), dtype=torch.float64, device="cuda")
        err = torch.empty((self.n_channels, cap), dtype=torch.float64, device="cuda")
        cnt = torch.zeros(self.n_channels, dtype=torch.int32, device="cuda")
        stride = x_dev.stride(0) if self.n_channels > 1 else n
        _lib.check(_lib.lib.wh_gardner_bank_run(self._h, x_dev.data_ptr(), n, stride, sym.data_ptr(), err.data_ptr(),
                                                cap, cnt.data_ptr(), _lib.stream_ptr(torch)), "wh_gardner_bank_run")
        return sym, err, cnt


class GardnerTED:
    """Single-channel drop-in (symbol_timing.py:60-211)."""

    def __init__(self, samples_per_symbol: float, loop_bw: float = 0.01, damping: float = 1.0):
        self.samples_per_symbol = samples_per_symbol
        self._bank = GardnerBank(1, samples_per_symbol, loop_bw, damping)

    def reset(self) -> None:
        self._bank.reset()

    def process_block(self, samples):
        x = np.ascontiguousarray(samples, dtype=np.float32)
        if x.size == 0:
            return np.array([], dtype=np.float64), np.array([], dtype=np.float64)
        torch = self._bank._torch
        s, e, c = self._bank.process_device(torch.from_numpy(x[None, :]).cuda())
        k = int(c[0])
        return s[0, :k].cpu().numpy(), e[0, :k].cpu().numpy()


class CostasBank:
    """CostasLoop (cqpsk.py:84-196) for C channels at once: complex128 [C, n] in -> phase-corrected complex128 [C, n]."""

    def __init__(self, n_channels: int, loop_bw: float = 0.01, damping: float = 0.707, max_freq: float = 0.1):
        self._torch = _lib.require_gpu()
        self.n_channels = int(n_channels)
        self._kp, self._ki = calculate_loop_coefficients(0.0, loop_bw, damping)     # cqpsk.py:107-110
        self._max_freq = max_freq
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_costas_bank_destroy
        _lib.check(_lib.lib.wh_costas_bank_create(C.byref(self._h), self.n_channels, self._kp, self._ki, float(max_freq)),
                   "wh_costas_bank_create")

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self) -> None:
        _lib.check(_lib.lib.wh_costas_bank_reset(self._h, _lib.stream_ptr(self._torch)), "wh_costas_bank_reset")

    def process_device(self, x_dev):
        torch = self._torch
        assert x_dev.is_cuda and x_dev.dtype == torch.complex128 and x_dev.dim() == 2 and x_dev.shape[0] == self.n_channels
        x_dev = x_dev.contiguous()
        out = torch.empty_like(x_dev)
        _lib.check(_lib.lib.wh_costas_bank_run(self._h, x_dev.data_ptr(), x_dev.shape[1], x_dev.shape[1], out.data_ptr(),
                                               None, _lib.stream_ptr(torch)), "wh_costas_bank_run")
        return out

    def frequency_offsets(self) -> np.ndarray:
        f = np.zeros(self.n_channels, dtype=np.float64)
        _lib.check(_lib.lib.wh_costas_bank_run(self._h, None, 0, 0, None, f.ctypes.data, _lib.stream_ptr(self._torch)),
                   "wh_costas_bank_run")
        return f


class CostasLoop:
    """Single-channel drop-in for cqpsk.py:84-196: `process(sample)`, `process_block(samples) -> complex128`,
    `reset()`, `frequency_offset`."""

    def __init__(self, loop_bw: float = 0.01, damping: float = 0.707, max_freq: float = 0.1):
        self._bank = CostasBank(1, loop_bw, damping, max_freq)
        self._kp, self._ki, self._max_freq = self._bank._kp, self._bank._ki, max_freq

    def reset(self) -> None:
        self._bank.reset()

    def process_block(self, samples):
        x = np.ascontiguousarray(samples, dtype=np.complex128)
        if x.size == 0:
            return np.zeros(0, dtype=np.complex128)
        torch = self._bank._torch
        return self._bank.process_device(torch.from_numpy(x[None, :]).cuda())[0].cpu().numpy()

    def process(self, sample: complex) -> complex:
        return complex(self.process_block(np.array([sample], dtype=np.complex128))[0])

    @property
    def frequency_offset(self) -> float:
        return float(self._bank.frequency_offsets()[0])


class MuellerMullerBank:
    """MuellerMullerTED (symbol_timing.py:214-380) for C channels: complex128 [C, n] -> (symbols, decisions complex128
    [C, cap], errors float64 [C, cap], counts int32 [C]) on the GPU."""

    def __init__(self, n_channels: int, samples_per_symbol: float, loop_bw: float = 0.01, damping: float = 1.0):
        self._torch = _lib.require_gpu()
        self.n_channels = int(n_channels)
        self.samples_per_symbol = samples_per_symbol
        self._kp, self._ki = calculate_loop_coefficients(samples_per_symbol, loop_bw, damping)
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_mm_bank_destroy
        _lib.check(_lib.lib.wh_mm_bank_create(C.byref(self._h), self.n_channels, float(samples_per_symbol), self._kp, self._ki),
                   "wh_mm_bank_create")

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self) -> None:
        _lib.check(_lib.lib.wh_mm_bank_reset(self._h, _lib.stream_ptr(self._torch)), "wh_mm_bank_reset")

    def process_device(self, x_dev):
        torch = self._torch
        assert x_dev.is_cuda and x_dev.dtype == torch.complex128 and x_dev.dim() == 2 and x_dev.shape[0] == self.n_channels
        x_dev = x_dev.contiguous()
        n = x_dev.shape[1]
        cap = int(n / (self.samples_per_symbol * 0.5)) + 8     # >= the library's bound n / (sps / 2) + 2, in floating point
        sym = torch.empty((self.n_channels, cap), dtype=torch.complex128, device=x_dev.device)
        dec = torch.empty((self.n_channels, cap), dtype=torch.complex128, device=x_dev.device)
        err = torch.empty((self.n_channels, cap), dtype=torch.float64, device=x_dev.device)
        cnt = torch.zeros(self.n_channels, dtype=torch.int32, device=x_dev.device)
        _lib.check(_lib.lib.wh_mm_bank_run(self._h, x_dev.data_ptr(), n, n, sym.data_ptr(), dec.data_ptr(), err.data_ptr(), cap,
                                           cnt.data_ptr(), _lib.stream_ptr(torch)), "wh_mm_bank_run")
        return sym, dec, err, cnt


class MuellerMullerTED:
    """Single-channel drop-in for symbol_timing.py:214-380: `process_block(samples) -> (symbols, decisions, errors)`."""

    def __init__(self, samples_per_symbol: float, loop_bw: float = 0.01, damping: float = 1.0):
        self.samples_per_symbol = samples_per_symbol
        self._bank = MuellerMullerBank(1, samples_per_symbol, loop_bw, damping)
        self._kp, self._ki = self._bank._kp, self._bank._ki

    def reset(self) -> None:
        self._bank.reset()

    def process_block(self, samples):
        x = np.ascontiguousarray(samples, dtype=np.complex128)
        if x.size == 0:
            return np.zeros(0, np.complex128), np.zeros(0, np.complex128), np.zeros(0, np.float64)
        torch = self._bank._torch
        s, d, e, c = self._bank.process_device(torch.from_numpy(x[None, :]).cuda())
        k = int(c[0])
        return s[0, :k].cpu().numpy(), d[0, :k].cpu().numpy(), e[0, :k].cpu().numpy()
