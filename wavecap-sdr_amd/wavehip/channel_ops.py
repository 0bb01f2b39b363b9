"""Channel operator: the batched MI355X form of capture.py:298-439
`_process_channel_dsp_stateless(samples, sample_rate, cfg) -> (audio | None, metrics)`.

`ChannelBank` runs all channels of one capture (and any number of consecutive
chunks) in one launch: the IQ chunk is shared by every channel through L2, the NCO mix
(capture.py:166-193), FM discriminator (dsp/fm.py:65-97), RMS normalise (fm.py:42-62),
resample_poly (fm.py:184-221) and soft clip (fm.py:26-39) are fused on the device and the
two metrics the lifecycle consumes (`rssi_db`, `signal_power_db`, capture.py:2579-2582)
come back with the audio.

`process_channel_dsp_stateless` is the single-channel drop-in with the reference's
exact signature and error convention (non-finite input or audio failing
validate_audio_samples -> (None, metrics); capture.py:323-325, 433-435).
Supported modes: "nbfm", "wbfm" (mode defaults of capture.py:3425-3496).  Other
modes raise NotImplementedError -- the integration stub keeps routing them to the
reference implementation; nothing here falls back to a CPU path.
"""

from __future__ import annotations

import ctypes as C
import logging
from dataclasses import dataclass, field
from math import gcd
from typing import Any

import numpy as np
from scipy import signal

from . import _lib

logger = logging.getLogger(__name__)

AUDIO_MAX_ABS = 1.2  # validation.py:9


@dataclass
class ChannelConfig:
    """Subset mirror of capture.py:442-501 (same field names; the reference's own
    ChannelConfig instances are accepted as-is, this class exists for standalone use)."""

    id: str = "ch"
    capture_id: str = "cap"
    mode: str = "nbfm"
    offset_hz: float = 0.0
    audio_rate: int = 48_000
    enable_deemphasis: bool = True
    deemphasis_tau_us: float = 75.0
    enable_mpx_filter: bool = True
    mpx_cutoff_hz: float = 15_000
    enable_fm_highpass: bool = False
    enable_fm_lowpass: bool = False
    notch_frequencies: list = field(default_factory=list)
    enable_noise_reduction: bool = False


def resample_design(in_rate: int, out_rate: int):
    """Taps/alignment of scipy.signal.resample_poly as called by dsp/fm.py:184-221:
    y[m] = sum_j h[j] * xup[m*down + d0 - j], h = firwin(20*max(up,down)+1, 1/max, kaiser 5.0)*up."""
    g = gcd(int(in_rate), int(out_rate))
    up, down = int(out_rate) // g, int(in_rate) // g
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = signal.firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float64) * up
    return np.ascontiguousarray(h), up, down, half_len


def _unsupported(cfg) -> str | None:
    mode = cfg.mode
    if mode not in ("nbfm", "wbfm"):
        return f"mode {mode!r}"
    if getattr(cfg, "notch_frequencies", None):
        return "notch filters"
    if getattr(cfg, "enable_noise_reduction", False):
        return "spectral noise reduction"
    if getattr(cfg, "enable_fm_highpass", False) or getattr(cfg, "enable_fm_lowpass", False):
        return "optional FM high/low-pass"
    if mode == "nbfm" and getattr(cfg, "enable_deemphasis", False):
        return "NBFM de-emphasis"
    return None


class ChannelBank:
    """All channels of one capture, one mode, one audio rate."""

    def __init__(self, sample_rate: int, chunk_len: int, cfgs: list, input_format: str = "cf32"):
        if not cfgs:
            raise ValueError("ChannelBank needs at least one channel")
        for c in cfgs:
            why = _unsupported(c)
            if why:
                raise NotImplementedError(f"wavehip ChannelBank: {why} is not implemented on the device")
        modes = {c.mode for c in cfgs}
        rates = {int(c.audio_rate) for c in cfgs}
        if len(modes) != 1 or len(rates) != 1:
            raise ValueError("ChannelBank: all channels must share mode and audio_rate (group them first)")
        self.mode = modes.pop()
        self.audio_rate = rates.pop()
        self.sample_rate = int(sample_rate)
        self.chunk_len = int(chunk_len)
        self.cfgs = list(cfgs)
        self.K = len(cfgs)
        self.input_format = {"cf32": 0, "int16": 1}[input_format]
        self._torch = _lib.require_gpu()
        if self.sample_rate == self.audio_rate:
            raise NotImplementedError("ChannelBank: sample_rate == audio_rate (no resampling) is not implemented")
        h, up, down, d0 = resample_design(self.sample_rate, self.audio_rate)
        self.up, self.down = up, down
        n_up = self.chunk_len * up
        self.n_out = n_up // down + (1 if n_up % down else 0)
        offs = np.array([int(round(float(c.offset_hz))) if float(c.offset_hz) != 0.0 else 0 for c in cfgs],
                        dtype=np.int32)
        cfg = _lib.FmBankCfg()
        cfg.sample_rate, cfg.chunk_len, cfg.n_channels = self.sample_rate, self.chunk_len, self.K
        cfg.h_offsets_hz = _lib.dptr(offs, "i32")
        cfg.input_format = self.input_format
        cfg.mode = 0 if self.mode == "nbfm" else 1
        cfg.h_taps = _lib.dptr(h, "f64")
        cfg.ntaps, cfg.up, cfg.down, cfg.d0, cfg.n_out = len(h), up, down, d0, self.n_out
        keep = [offs, h]
        cfg.deemph_b0, cfg.deemph_a1 = 0.0, 0.0
        if self.mode == "wbfm":
            c0 = cfgs[0]
            for c in cfgs:
                if (c.enable_deemphasis, c.deemphasis_tau_us, c.enable_mpx_filter, c.mpx_cutoff_hz) != \
                        (c0.enable_deemphasis, c0.deemphasis_tau_us, c0.enable_mpx_filter, c0.mpx_cutoff_hz):
                    raise ValueError("ChannelBank(wbfm): channels must share the de-emphasis / MPX settings")
            if c0.enable_deemphasis:  # dsp/fm.py:101-108 (tau quantised to integer microseconds)
                tau = int((c0.deemphasis_tau_us * 1e-6) * 1e6) * 1e-6
                alpha = 1.0 / (1.0 + (1.0 / (2.0 * np.pi * tau * self.sample_rate)))
                cfg.deemph_b0 = float(np.float32(alpha))
                cfg.deemph_a1 = float(np.float32(-(1.0 - alpha)))
            if c0.enable_mpx_filter:  # dsp/fm.py:129-144
                nc = int(c0.mpx_cutoff_hz) / (self.sample_rate / 2.0)
                if nc < 1.0:
                    b, a = signal.butter(5, nc, btype="low")
                    b = np.ascontiguousarray(b, dtype=np.float64)
                    a = np.ascontiguousarray(a, dtype=np.float64)
                    cfg.h_mpx_b, cfg.h_mpx_a = _lib.dptr(b, "f64"), _lib.dptr(a, "f64")
                    keep += [b, a]
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_fmbank_destroy
        _lib.check(_lib.lib.wh_fmbank_create(C.byref(self._h), C.byref(cfg)), "wh_fmbank_create")
        del keep

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def process_device(self, d_in, n_chunks: int, audio=None, metrics=None):
        """d_in: GPU tensor holding n_chunks*chunk_len samples (complex64, or int16 pairs).
        Returns (audio f32 [n_chunks, K, n_out], metrics f32 [n_chunks, K, 4]) on the GPU."""
        torch = self._torch
        assert d_in.is_cuda and d_in.is_contiguous()
        if self.input_format == 0:
            assert d_in.dtype == torch.complex64 and d_in.numel() >= n_chunks * self.chunk_len
        else:
            assert d_in.dtype == torch.int16 and d_in.numel() >= 2 * n_chunks * self.chunk_len
        if audio is None:
            audio = torch.empty((n_chunks, self.K, self.n_out), dtype=torch.float32, device=d_in.device)
        if metrics is None:
            metrics = torch.empty((n_chunks, self.K, 4), dtype=torch.float32, device=d_in.device)
        _lib.check(_lib.lib.wh_fmbank_run(self._h, d_in.data_ptr(), n_chunks, audio.data_ptr(), metrics.data_ptr(),
                                          _lib.stream_ptr(torch)), "wh_fmbank_run")
        return audio, metrics

    def process(self, samples) -> list[tuple[np.ndarray | None, dict[str, Any]]]:
        """One chunk (host array) -> [(audio | None, metrics)] per channel, reference conventions."""
        torch = self._torch
        if self.input_format == 0:
            x = np.ascontiguousarray(samples, dtype=np.complex64)
            n = x.shape[0]
        else:
            x = np.ascontiguousarray(samples, dtype=np.int16)
            n = x.shape[0] // 2
        if n != self.chunk_len:
            raise ValueError(f"ChannelBank.process: expected {self.chunk_len} samples, got {n}")
        if self.input_format == 0 and not np.isfinite(x.view(np.float32)).all():  # capture.py:323-325
            logger.warning("ChannelBank: non-finite IQ samples, dropping DSP chunk")
            return [(None, {}) for _ in range(self.K)]
        audio, met = self.process_device(torch.from_numpy(x).cuda(), 1)
        audio, met = audio[0].cpu().numpy(), met[0].cpu().numpy()
        out = []
        for k in range(self.K):
            m: dict[str, Any] = {"rssi_db": float(met[k, 0])}
            if met[k, 3] < 0.5 or met[k, 2] > AUDIO_MAX_ABS:  # validation.py:41-52
                out.append((None, m))
                continue
            m["signal_power_db"] = float(met[k, 1])
            out.append((audio[k].copy(), m))
        return out


_bank_cache: dict[tuple, ChannelBank] = {}


def process_channel_dsp_stateless(samples, sample_rate: int, cfg) -> tuple[np.ndarray | None, dict[str, Any]]:
    """Drop-in for capture.py:298 (single channel).  Banks are cached per
    (rate, chunk length, mode, offset, audio rate, filter settings)."""
    metrics: dict[str, Any] = {}
    if samples.size == 0:
        return None, metrics
    why = _unsupported(cfg)
    if why:
        raise NotImplementedError(f"wavehip: {why} is not implemented on the device")
    key = (int(sample_rate), int(samples.shape[0]), cfg.mode, int(round(float(cfg.offset_hz))), int(cfg.audio_rate),
           bool(cfg.enable_deemphasis), float(cfg.deemphasis_tau_us), bool(cfg.enable_mpx_filter),
           float(cfg.mpx_cutoff_hz))
    bank = _bank_cache.get(key)
    if bank is None:
        if len(_bank_cache) >= 64:
            _bank_cache.pop(next(iter(_bank_cache)))
        bank = _bank_cache[key] = ChannelBank(sample_rate, samples.shape[0], [cfg])
    return bank.process(samples)[0]
