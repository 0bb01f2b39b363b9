"""Channel operator: the batched MI355X form of capture.py:298-439
`_process_channel_dsp_stateless(samples, sample_rate, cfg) -> (audio | None, metrics)`.

`ChannelBank` runs all channels of one capture (and any number of consecutive chunks) in one
launch: the IQ chunk is shared by every channel through L2; NCO mix (capture.py:166-193), the
demodulator front (FM discriminator dsp/fm.py:65-97, AM envelope dsp/am.py:103, SSB product
detector dsp/am.py:200-210), the optional IIR stages (de-emphasis, MPX / Butterworth / notch,
dsp/fm.py:101-181, dsp/filters.py:86-264), AGC (dsp/agc.py:169-242), resample_poly
(dsp/fm.py:184-221) and the soft clip run on the device, and the two metrics the lifecycle
consumes (`rssi_db`, `signal_power_db`, capture.py:2579-2582) come back with the audio.  Every
filter is designed here with the very scipy calls the reference makes, so coefficients are
identical.

`process_channel_dsp_stateless` is the single-channel drop-in with the reference's exact
signature and error convention (non-finite input or audio failing validate_audio_samples ->
(None, metrics); capture.py:323-325, 433-435).  Supported modes: "nbfm", "wbfm", "am", "ssb", "sam"
with their filter flags, "raw" (mixed IQ, interleaved) and the digital voice modes (metrics only,
capture.py:424-430).  Unknown modes raise NotImplementedError -- the integration stub keeps routing those to the reference implementation
(INTEGRATION.md); nothing here falls back to a CPU path.
"""

from __future__ import annotations

import ctypes as C
import logging
import threading
from dataclasses import dataclass, field
from math import gcd
from typing import Any

import numpy as np
from scipy import signal

from . import _lib

logger = logging.getLogger(__name__)

AUDIO_MAX_ABS = 1.2  # validation.py:9
NR_FFT_SIZE = 1024   # dsp/filters.py:350


@dataclass
class ChannelConfig:
    """Mirror of capture.py:440-501 (same field names and defaults -- the build-container boundary check compares them with the
    reference's dataclass; the reference's own ChannelConfig instances are accepted as-is, this class exists for
    standalone use, so `id` / `capture_id` have defaults here).  Like the reference's dispatcher (capture.py:340-414)
    the operator does not forward agc_attack_ms / agc_release_ms / enable_noise_blanker; name, RDS and POCSAG fields
    belong to the lifecycle around the operator."""

    id: str = "ch"
    capture_id: str = "cap"
    mode: str = "nbfm"
    offset_hz: float = 0.0
    audio_rate: int = 48_000
    name: str | None = None
    auto_name: str | None = None
    enable_deemphasis: bool = True
    deemphasis_tau_us: float = 75.0
    enable_mpx_filter: bool = True
    mpx_cutoff_hz: float = 15_000
    enable_fm_highpass: bool = False
    fm_highpass_hz: float = 100
    enable_fm_lowpass: bool = False
    fm_lowpass_hz: float = 3_000
    enable_am_highpass: bool = True
    am_highpass_hz: float = 100
    enable_am_lowpass: bool = True
    am_lowpass_hz: float = 5_000
    enable_ssb_bandpass: bool = True
    ssb_bandpass_low_hz: float = 300
    ssb_bandpass_high_hz: float = 3_000
    ssb_mode: str = "usb"
    ssb_bfo_offset_hz: float = 1500.0
    enable_agc: bool = False
    agc_target_db: float = -20.0
    agc_attack_ms: float = 5.0
    agc_release_ms: float = 50.0
    enable_noise_blanker: bool = False
    noise_blanker_threshold_db: float = 10.0
    notch_frequencies: list = field(default_factory=list)
    enable_noise_reduction: bool = False
    noise_reduction_db: float = 12.0
    squelch_db: float | None = None
    sam_sideband: str = "dsb"
    sam_pll_bandwidth_hz: float = 50.0
    enable_rds: bool = True
    enable_pocsag: bool = False
    pocsag_baud: int = 1200


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
    if cfg.mode not in ("nbfm", "wbfm", "am", "ssb", "sam"):
        return f"mode {cfg.mode!r}"
    return None


def _stage(b, a, is_f64: bool):
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    n = max(len(b), len(a))
    if n > 11:
        raise NotImplementedError("IIR section longer than 11 coefficients")
    st = _lib.IirStage()
    st.is_f64, st.n = (1 if is_f64 else 0), n
    for k in range(11):
        st.b[k] = float(b[k]) if k < len(b) else 0.0
        st.a[k] = float(a[k]) if k < len(a) else 0.0
    return st


def _butter(kind: str, sample_rate: int, cutoff):
    """dsp/filters.py:86-221 validity rules + cached butter(5) design."""
    nyq = sample_rate / 2.0
    if kind == "band":
        lo, hi = cutoff[0] / nyq, cutoff[1] / nyq
        if lo <= 0 or hi >= 1.0 or lo >= hi:
            return None
        return signal.butter(5, [lo, hi], btype="band")
    wn = cutoff / nyq
    if wn <= 0 or wn >= 1.0:
        return None
    return signal.butter(5, wn, btype=kind)


def _notches(cfg, sample_rate: int):
    out = []
    for f in getattr(cfg, "notch_frequencies", None) or []:
        if 0 < f < sample_rate / 2:                     # dsp/fm.py:297-300
            nf = f / (sample_rate / 2.0)
            if 0 < nf < 1.0:                            # dsp/filters.py:252-254
                out.append(_stage(*signal.iirnotch(nf, 30.0), True))
    return out


def build_chain(cfg, sample_rate: int):
    """-> (demod, bfo_hz, stages, agc tuple | None, post) in the reference's application order
    (SAM's PLL coefficients come from sam_pll_coefficients)."""
    mode = cfg.mode
    stages = []
    agc = None
    bfo = 0.0
    if mode in ("nbfm", "wbfm"):
        demod, post = 0, 0
        if cfg.enable_deemphasis:                       # dsp/fm.py:101-126 (tau as integer microseconds)
            tau = int((cfg.deemphasis_tau_us * 1e-6) * 1e6) * 1e-6
            alpha = 1.0 / (1.0 + (1.0 / (2.0 * np.pi * tau * sample_rate)))
            stages.append(_stage(np.array([alpha], np.float32), np.array([1.0, -(1.0 - alpha)], np.float32), False))
        if mode == "wbfm" and cfg.enable_mpx_filter:    # dsp/fm.py:129-181
            nc = int(cfg.mpx_cutoff_hz) / (sample_rate / 2.0)
            if nc < 1.0:
                stages.append(_stage(*signal.butter(5, nc, btype="low"), True))
        if cfg.enable_fm_highpass and cfg.fm_highpass_hz > 0:
            ba = _butter("high", sample_rate, cfg.fm_highpass_hz)
            if ba is not None:
                stages.append(_stage(*ba, True))
        if mode == "nbfm" and cfg.enable_fm_lowpass and cfg.fm_lowpass_hz > 0:
            ba = _butter("low", sample_rate, cfg.fm_lowpass_hz)
            if ba is not None:
                stages.append(_stage(*ba, True))
        stages += _notches(cfg, sample_rate)
    else:
        post = 1
        if mode in ("am", "sam"):                       # dsp/am.py:45-141; dsp/sam.py:132-269 (same chain after the front)
            demod = 1 if mode == "am" else {"usb": 4, "lsb": 5}.get(str(getattr(cfg, "sam_sideband", "dsb")).lower(), 3)
            if cfg.enable_am_highpass and cfg.am_highpass_hz > 0:
                ba = _butter("high", sample_rate, cfg.am_highpass_hz)
                if ba is not None:
                    stages.append(_stage(*ba, True))
            if cfg.enable_am_lowpass and cfg.am_lowpass_hz > 0:
                ba = _butter("low", sample_rate, cfg.am_lowpass_hz)
                if ba is not None:
                    stages.append(_stage(*ba, True))
        else:                                           # dsp/am.py:144-247
            demod = 2
            bfo = cfg.ssb_bfo_offset_hz if cfg.ssb_mode.lower() == "usb" else -cfg.ssb_bfo_offset_hz
            if cfg.enable_ssb_bandpass:
                ba = _butter("band", sample_rate, (cfg.ssb_bandpass_low_hz, cfg.ssb_bandpass_high_hz))
                if ba is not None:
                    stages.append(_stage(*ba, True))
        if mode != "sam":                               # capture.py:386-399 passes no notch list to sam_demod_simple
            stages += _notches(cfg, sample_rate)
        if cfg.enable_agc:                              # dsp/agc.py:169-242, attack 5 ms / release 50 ms
            att = (5.0 / 1000.0) * sample_rate
            rel = (50.0 / 1000.0) * sample_rate
            ac = 1.0 - np.exp(-1.0 / att) if att > 0 else 1.0
            rc = 1.0 - np.exp(-1.0 / rel) if rel > 0 else 1.0
            agc = (np.float32(10.0 ** (cfg.agc_target_db / 20.0)), np.float32(10.0 ** (60.0 / 20.0)),
                   np.float32(ac), np.float32(-(1.0 - ac)), np.float32(rc), np.float32(-(1.0 - rc)))
    if len(stages) > 8:
        raise NotImplementedError("more than 8 IIR sections in one channel")
    return demod, float(bfo), stages, agc, post


def iir_warmup_samples(stages, tol: float = 1e-10) -> int:
    """Samples after which the impulse response of every stage of the cascade is below `tol` (relative; the outputs
    are float32, 6e-8, and the parity bar is 1e-5): from the
    largest pole radius rho, n = log(tol) / log(rho), padded for the polynomial factor of clustered poles.  0 when a
    pole sits on / outside the unit circle or the answer is useless (> 2^22)."""
    rho, order = 0.0, 0
    for st in stages:
        a = np.array([st.a[k] for k in range(st.n)], dtype=np.float64)
        while a.size > 1 and a[-1] == 0.0:
            a = a[:-1]
        if a.size > 1:
            rho = max(rho, float(np.max(np.abs(np.roots(a)))))
            order += a.size - 1
    if order == 0:
        return 1
    if rho >= 1.0:
        return 0
    if rho < 1e-6:
        return order + 1
    n = np.log(tol) / np.log(rho)
    n = 1.25 * n + 32 * order
    return int(n) if n < (1 << 22) else 0


def iir_scan_safe(stages, segment: int, limit: float = 64.0) -> bool:
    """True when every stage's DF2T transition matrix raised to `segment` has entries <= `limit`: then the start
    states of the scan form (s' = M s + e) carry no cancellation.  The reference's order-5 ba-form high-/band-passes
    fail this by many orders of magnitude (entries ~1e6 at 48 kS/s), low-passes, notches and one-poles pass."""
    for st in stages:
        m = st.n - 1
        if m < 1:
            continue
        a = np.array([st.a[k] for k in range(st.n)], dtype=np.float64) / st.a[0]
        A = np.zeros((m, m))
        A[:, 0] = -a[1:]
        A[np.arange(m - 1), np.arange(1, m)] += 1.0
        with np.errstate(over="ignore", invalid="ignore"):
            M = np.linalg.matrix_power(A, int(segment))
        if not np.isfinite(M).all() or np.abs(M).max() > limit:
            return False
    return True


def sam_pll_coefficients(sample_rate: int, loop_bandwidth: float, damping: float = 0.707):
    """dsp/sam.py:55-66 (sample_rate as float, Python-float arithmetic in the reference's order)."""
    fs = float(sample_rate)
    omega_n = 2 * np.pi * loop_bandwidth
    return float(2 * damping * omega_n / fs), float((omega_n ** 2) / (fs ** 2))


def _chain_key(cfg):
    return (cfg.mode, int(cfg.audio_rate), bool(cfg.enable_deemphasis), float(cfg.deemphasis_tau_us),
            bool(cfg.enable_mpx_filter), float(cfg.mpx_cutoff_hz), bool(cfg.enable_fm_highpass),
            float(cfg.fm_highpass_hz), bool(cfg.enable_fm_lowpass), float(cfg.fm_lowpass_hz),
            bool(cfg.enable_am_highpass), float(cfg.am_highpass_hz), bool(cfg.enable_am_lowpass),
            float(cfg.am_lowpass_hz), bool(cfg.enable_ssb_bandpass), float(cfg.ssb_bandpass_low_hz),
            float(cfg.ssb_bandpass_high_hz), str(cfg.ssb_mode), float(cfg.ssb_bfo_offset_hz),
            bool(cfg.enable_agc), float(cfg.agc_target_db), tuple(getattr(cfg, "notch_frequencies", None) or ()),
            str(getattr(cfg, "sam_sideband", "dsb")).lower(), float(getattr(cfg, "sam_pll_bandwidth_hz", 50.0)),
            bool(getattr(cfg, "enable_noise_reduction", False)), float(getattr(cfg, "noise_reduction_db", 12.0)))


def _offsets_i32(cfgs) -> np.ndarray:
    """round(offset_hz) per channel; exactly 0.0 means "no mix" (capture.py:326-329)."""
    return np.array([int(round(float(c.offset_hz))) if float(c.offset_hz) != 0.0 else 0 for c in cfgs], dtype=np.int32)


def _squelch_key(cfgs) -> tuple:
    return tuple(None if getattr(c, "squelch_db", None) is None else float(c.squelch_db) for c in cfgs)


class ChannelBank:
    """All channels of one capture that share one chain (mode + filter settings + audio rate)."""

    def __init__(self, sample_rate: int, chunk_len: int, cfgs: list, input_format: str = "cf32",
                 apply_squelch: bool = False, iir_form: str = "auto"):
        """apply_squelch=True additionally zeroes the audio of chunks whose rssi_db is below the channel's
        squelch_db (capture.py:2918-2921; in the reference that happens later, in _apply_stateful_processing,
        so the drop-in for _process_channel_dsp_stateless leaves it off).  iir_form="sequential" keeps the IIR stages
        on the one-lane-per-row recurrence (tests compare the parallel forms against it); "warmup_recurrence" makes
        the time-parallel warm-up form run the recurrences over its warm-up samples instead of taking the start
        states from the chain's impulse response (the older, 16x more expensive way to the same states)."""
        if not cfgs:
            raise ValueError("ChannelBank needs at least one channel")
        for c in cfgs:
            why = _unsupported(c)
            if why:
                raise NotImplementedError(f"wavehip ChannelBank: {why} is not implemented on the device")
        if len({_chain_key(c) for c in cfgs}) != 1:
            raise ValueError("ChannelBank: all channels must share mode, filter settings and audio_rate "
                             "(group them first)")
        c0 = cfgs[0]
        self.mode = c0.mode
        self.audio_rate = int(c0.audio_rate)
        self.sample_rate = int(sample_rate)
        self.chunk_len = int(chunk_len)
        self.cfgs = list(cfgs)
        self.K = len(cfgs)
        self.input_format = {"cf32": 0, "int16": 1}[input_format]
        self._torch = _lib.require_gpu()
        demod, bfo, stages, agc, post = build_chain(c0, self.sample_rate)
        offs = _offsets_i32(cfgs)
        self.offsets = tuple(int(v) for v in offs)
        self.squelch = None
        cfg = _lib.ChanBankCfg()
        cfg.sample_rate, cfg.chunk_len, cfg.n_channels = self.sample_rate, self.chunk_len, self.K
        cfg.h_offsets_hz = _lib.dptr(offs, "i32")
        cfg.input_format = self.input_format
        cfg.demod, cfg.bfo_hz, cfg.post = demod, bfo, post
        if demod >= 3:
            cfg.pll_alpha, cfg.pll_beta = sam_pll_coefficients(self.sample_rate, float(getattr(c0, "sam_pll_bandwidth_hz", 50.0)))
        keep = [offs]
        cfg.n_stages = len(stages)
        if iir_form not in ("auto", "sequential", "warmup_recurrence"):
            raise ValueError("iir_form must be 'auto', 'sequential' or 'warmup_recurrence'")
        par = iir_form != "sequential"
        cfg.iir_warmup_form = 1 if iir_form == "warmup_recurrence" else 0
        cfg.iir_warmup = iir_warmup_samples(stages) if (par and stages and agc is None and demod < 3) else 0
        # (the library cuts a row into 64, 256 or 512 segments by channel count: safe for each of them)
        cfg.iir_scan = 1 if (par and all(iir_scan_safe(stages, (self.chunk_len + 64 * w - 1) // (64 * w)) for w in (1, 4, 8))) else 0
        if stages:
            arr = (_lib.IirStage * len(stages))(*stages)
            cfg.h_stages = arr
            keep.append(arr)
        cfg.agc = 1 if agc is not None else 0
        if agc is not None:
            (cfg.agc_target, cfg.agc_max_gain, cfg.agc_att_b0, cfg.agc_att_a1, cfg.agc_rel_b0,
             cfg.agc_rel_a1) = (float(v) for v in agc)
        if apply_squelch and any(getattr(c, "squelch_db", None) is not None for c in cfgs):
            sq = np.array([np.nan if getattr(c, "squelch_db", None) is None else float(c.squelch_db) for c in cfgs],
                          dtype=np.float32)
            cfg.h_squelch_db = _lib.dptr(sq, "f32")
            keep.append(sq)
            self.squelch = _squelch_key(cfgs)
        n_fm = self.chunk_len
        if self.mode in ("nbfm", "wbfm") and getattr(c0, "enable_noise_reduction", False):
            # dsp/filters.py:346-460 between the filters and rms_normalize (dsp/fm.py:303-304, 399-400); the row
            # comes back (n_frames-1)*512 + 1024 samples long
            cfg.noise_reduction = 1
            cfg.nr_reduction_linear = float(np.float32(10 ** (float(getattr(c0, "noise_reduction_db", 12.0)) / 20.0)))
            win = np.ascontiguousarray(signal.windows.hann(NR_FFT_SIZE, sym=False).astype(np.float32))
            cfg.h_nr_window = _lib.dptr(win, "f32")
            keep.append(win)
            if n_fm >= NR_FFT_SIZE:
                n_fm = ((n_fm - NR_FFT_SIZE) // (NR_FFT_SIZE // 2)) * (NR_FFT_SIZE // 2) + NR_FFT_SIZE
        self.n_fm = n_fm
        if self.sample_rate == self.audio_rate:         # dsp/fm.py:198-199: no resampling
            self.up = self.down = 1
            self.n_out = n_fm
            cfg.ntaps = 0
            cfg.up = cfg.down = 1
            cfg.d0 = 0
        else:
            h, up, down, d0 = resample_design(self.sample_rate, self.audio_rate)
            keep.append(h)
            self.up, self.down = up, down
            n_up = n_fm * up
            self.n_out = n_up // down + (1 if n_up % down else 0)
            cfg.h_taps = _lib.dptr(h, "f64")
            cfg.ntaps, cfg.up, cfg.down, cfg.d0 = len(h), up, down, d0
        cfg.n_out = self.n_out
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_chanbank_destroy
        _lib.check(_lib.lib.wh_chanbank_create(C.byref(self._h), C.byref(cfg)), "wh_chanbank_create")
        del keep

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def set_offsets(self, offsets_hz) -> None:
        """Retune the bank's channels (API PATCH of offset_hz -> capture.py:442-501; the reference's operator reads
        cfg.offset_hz per chunk, capture.py:326-329): one 4 K-byte copy enqueued on the current stream, nothing is rebuilt.
        Rounding and the `offset_hz == 0.0 -> no mix` rule are the constructor's."""
        offs = np.array([int(round(float(o))) if float(o) != 0.0 else 0 for o in offsets_hz], dtype=np.int32)
        if offs.shape[0] != self.K:
            raise ValueError(f"ChannelBank.set_offsets: the bank has {self.K} channels")
        _lib.check(_lib.lib.wh_chanbank_set_offsets(self._h, offs.ctypes.data, self.K, _lib.stream_ptr(self._torch)),
                   "wh_chanbank_set_offsets")
        self.offsets = tuple(int(v) for v in offs)

    def set_squelch(self, squelch_db) -> None:
        """New squelch thresholds (None = no squelch for that channel) for a bank created with apply_squelch=True."""
        sq = np.array([np.nan if v is None else float(v) for v in squelch_db], dtype=np.float32)
        if sq.shape[0] != self.K:
            raise ValueError(f"ChannelBank.set_squelch: the bank has {self.K} channels")
        _lib.check(_lib.lib.wh_chanbank_set_squelch(self._h, sq.ctypes.data, self.K, _lib.stream_ptr(self._torch)),
                   "wh_chanbank_set_squelch")
        self.squelch = tuple(None if v is None else float(v) for v in squelch_db)

    WIRE = {None: 0, "pcm16": 1, "f32": 2}

    def process_device(self, d_in, n_chunks: int, audio=None, metrics=None, wire: str | None = None, wire_out=None):
        """d_in: GPU tensor holding n_chunks*chunk_len samples (complex64, or int16 pairs).
        Returns (audio f32 [n_chunks, K, n_out], metrics f32 [n_chunks, K, 4]) on the GPU; with wire="pcm16" / "f32"
        a third tensor [n_chunks, K, n_out] holds the audio in its wire format (int16 by the pack_pcm16 rule / float32
        clipped to [-1, 1], capture.py:119-144), written by the finalize kernel itself -- the only buffer that has
        to cross PCIe."""
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
        if wire is None:
            _lib.check(_lib.lib.wh_chanbank_run(self._h, d_in.data_ptr(), n_chunks, audio.data_ptr(), metrics.data_ptr(),
                                                _lib.stream_ptr(torch)), "wh_chanbank_run")
            return audio, metrics
        fmt = self.WIRE[wire]
        if wire_out is None:
            wire_out = torch.empty((n_chunks, self.K, self.n_out), dtype=torch.int16 if fmt == 1 else torch.float32,
                                   device=d_in.device)
        _lib.check(_lib.lib.wh_chanbank_run_wire(self._h, d_in.data_ptr(), n_chunks, audio.data_ptr(), metrics.data_ptr(),
                                                 fmt, wire_out.data_ptr(), _lib.stream_ptr(torch)), "wh_chanbank_run_wire")
        return audio, metrics, wire_out

    def process_wire(self, samples, wire: str = "pcm16") -> list[tuple[bytes | None, dict[str, Any]]]:
        """One chunk (host array) -> [(wire bytes | None, metrics)] per channel: what capture.py:884-948 hands to its
        audio sinks (pack_pcm16 / pack_f32 of the operator's audio), with only the wire buffer and the metrics
        downloaded.  Same validation conventions as process()."""
        torch = self._torch
        x = np.ascontiguousarray(samples, dtype=np.complex64 if self.input_format == 0 else np.int16)
        n = x.shape[0] // (1 if self.input_format == 0 else 2)
        if n != self.chunk_len:
            raise ValueError(f"ChannelBank.process_wire: expected {self.chunk_len} samples, got {n}")
        if self.input_format == 0 and not np.isfinite(x.view(np.float32)).all():  # capture.py:323-325
            logger.warning("ChannelBank: non-finite IQ samples, dropping DSP chunk")
            return [(None, {}) for _ in range(self.K)]
        _, met_dev, w_dev = self.process_device(torch.from_numpy(x).cuda(), 1, wire=wire)
        met, w = met_dev[0].cpu().numpy(), w_dev[0].cpu().numpy()
        out = []
        for k in range(self.K):
            m: dict[str, Any] = {"rssi_db": float(met[k, 0])}
            if met[k, 3] < 0.5 or met[k, 2] > AUDIO_MAX_ABS:  # validation.py:41-52
                out.append((None, m))
                continue
            m["signal_power_db"] = float(met[k, 1])
            out.append((w[k].tobytes(), m))
        return out

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
        return self.collect(*self.process_device(torch.from_numpy(x).cuda(), 1))

    def collect(self, audio_dev, met_dev) -> list[tuple[np.ndarray | None, dict[str, Any]]]:
        """Device results of ONE chunk -> the reference's (audio | None, metrics) per channel."""
        audio, met = audio_dev[0].cpu().numpy(), met_dev[0].cpu().numpy()
        out = []
        for k in range(self.K):
            m: dict[str, Any] = {"rssi_db": float(met[k, 0])}
            if met[k, 3] < 0.5 or met[k, 2] > AUDIO_MAX_ABS:  # validation.py:41-52
                out.append((None, m))
                continue
            m["signal_power_db"] = float(met[k, 1])
            out.append((audio[k].copy(), m))
        return out


def update_signal_metrics(samples, sample_rate: int, offsets_hz, input_format: str = "cf32",
                          shared_magnitudes: bool = False) -> list[dict[str, Any]]:
    """Channel.update_signal_metrics (capture.py:749-798) for all channels of one chunk in one pass:
    [{"rssi_db": float, "snr_db": float | None}] -- RSSI = 10 log10(mean |base|^2 + 1e-10); SNR from the
    10th / 90th percentile magnitudes exactly as np.partition picks them.  (The reference throttles the SNR
    part to every 10th call; here it costs one extra select pass, so it is always returned.)

    shared_magnitudes=True: the reference takes these statistics of `base = freq_shift(iq, offset)` BEFORE any
    filtering, and a unit-magnitude mix leaves |base| = |iq| up to float32 rounding (1e-7 relative, 1e-6 dB on the
    order statistics) -- so the chunk's magnitudes are ranked ONCE, without any mix, and every channel gets the same
    numbers: K channels cost what one costs (the ChannelDispatcher's snr option uses this)."""
    if shared_magnitudes:
        one = update_signal_metrics(samples, sample_rate, [0.0], input_format)[0]
        return [dict(one) for _ in offsets_hz]
    torch = _lib.require_gpu()
    if torch.is_tensor(samples):
        d = samples
        n = d.numel() // (2 if d.dtype == torch.int16 else 1)
        fmt = 1 if d.dtype == torch.int16 else 0
    else:
        fmt = {"cf32": 0, "int16": 1}[input_format]
        x = np.ascontiguousarray(samples, dtype=np.int16 if fmt else np.complex64)
        n = x.shape[0] // (2 if fmt else 1)
        d = torch.from_numpy(x).cuda()
    offs = np.array([int(round(float(o))) if float(o) != 0.0 else 0 for o in offsets_hz], dtype=np.int32)
    out = np.zeros((len(offs), 3), dtype=np.float32)
    _lib.check(_lib.lib.wh_channel_signal_metrics(d.data_ptr(), fmt, n, int(sample_rate), _lib.dptr(offs, "i32"),
                                                  len(offs), _lib.dptr(out, "f32"), _lib.stream_ptr(torch)),
               "wh_channel_signal_metrics")
    res = []
    k_noise, k_signal = n // 10, n - n // 10 - 1
    for k in range(len(offs)):
        m: dict[str, Any] = {"rssi_db": float(out[k, 0]), "snr_db": None}
        if k_noise > 0 and k_signal > k_noise:
            noise_power, signal_power = np.float32(out[k, 1]) ** 2, np.float32(out[k, 2]) ** 2
            if noise_power > 1e-10:
                m["snr_db"] = float(10.0 * np.log10(signal_power / noise_power))
        res.append(m)
    return res


_bank_cache: dict[tuple, tuple] = {}
_bank_cache_lock = threading.Lock()     # the reference calls the operator from a 3-thread pool (capture.py:1906-1925)

DIGITAL_MODES = ("p25", "dmr", "nxdn", "dstar", "ysf")      # capture.py:424


def _raw_or_digital(samples, sample_rate: int, cfg) -> tuple[np.ndarray | None, dict[str, Any]]:
    """capture.py:415-430: "raw" returns the frequency-shifted IQ interleaved as float32; the digital voice
    modes return no audio, only rssi_db / signal_power_db of the shifted IQ (their decoders are stateful and
    run elsewhere).  Mix (A2 kernel) and the power / validation reduction run on the device."""
    torch = _lib.require_gpu()
    metrics: dict[str, Any] = {}
    x = np.ascontiguousarray(samples, dtype=np.complex64)
    if not np.isfinite(x.view(np.float32)).all():          # capture.py:323-325
        logger.warning(f"Channel {getattr(cfg, 'id', '?')}: non-finite IQ samples, dropping DSP chunk")
        return None, metrics
    d = torch.from_numpy(x).cuda()
    off = float(cfg.offset_hz)
    if off != 0.0:
        base = torch.empty_like(d)
        _lib.check(_lib.lib.wh_nco_mix(d.data_ptr(), base.data_ptr(), x.shape[0], int(round(off)), int(sample_rate),
                                       _lib.stream_ptr(torch)), "wh_nco_mix")
    else:
        base = d
    st = (C.c_float * 3)()
    flat = torch.view_as_real(base)
    _lib.check(_lib.lib.wh_audio_stats(flat.data_ptr(), 2 * x.shape[0], st, _lib.stream_ptr(torch)), "wh_audio_stats")
    mean_sq, max_abs, finite = float(st[0]), float(st[1]), st[2] >= 0.5
    rssi = float(10.0 * np.log10(np.float32(2.0 * mean_sq) + np.float32(1e-10)))   # mean |base|^2 = 2 mean(re^2, im^2)
    metrics["rssi_db"] = rssi
    if cfg.mode in DIGITAL_MODES:
        metrics["signal_power_db"] = rssi
        return None, metrics
    if not finite or max_abs > AUDIO_MAX_ABS:              # validation.py:41-52
        return None, metrics
    metrics["signal_power_db"] = float(10.0 * np.log10(np.float32(mean_sq) + np.float32(1e-10)))
    return flat.reshape(-1).cpu().numpy(), metrics


def process_channel_dsp_stateless(samples, sample_rate: int, cfg) -> tuple[np.ndarray | None, dict[str, Any]]:
    """Drop-in for capture.py:298 (single channel).  Banks are cached per
    (rate, chunk length, offset, chain settings)."""
    metrics: dict[str, Any] = {}
    if samples.size == 0:
        return None, metrics
    if cfg.mode == "raw" or cfg.mode in DIGITAL_MODES:
        return _raw_or_digital(samples, sample_rate, cfg)
    why = _unsupported(cfg)
    if why:
        raise NotImplementedError(f"wavehip: {why} is not implemented on the device")
    key = (int(sample_rate), int(samples.shape[0]), int(round(float(cfg.offset_hz)))) + _chain_key(cfg)
    with _bank_cache_lock:
        entry = _bank_cache.get(key)
        if entry is None:
            if len(_bank_cache) >= 64:
                _bank_cache.pop(next(iter(_bank_cache)))      # the evicted bank lives on while a thread still holds it
            entry = _bank_cache[key] = (ChannelBank(sample_rate, samples.shape[0], [cfg]), threading.Lock())
    bank, lock = entry
    with lock:                                                # a bank owns one workspace: one call at a time
        return bank.process(samples)[0]


class ChannelDispatcher:
    """Batched replacement for the DSP fan-out of Capture._process_channels_parallel (capture.py:2489-2597, SURVEY
    8(f) N1): instead of one thread-pool task per running channel, the chunk is uploaded ONCE and every group of
    channels that shares a chain (mode + filter settings + audio rate) runs as one ChannelBank launch sequence;
    "raw" and the digital-voice modes (metrics only) ride along.  `process(samples, cfgs)` returns
    [(audio | None, metrics)] in the order of `cfgs`, with the conventions of _process_channel_dsp_stateless;
    the stateful tail (RDS / POCSAG / P25 decoders, audio metrics) stays with the caller, squelch can be fused
    (`apply_squelch=True`).  Banks are cached per (chunk length, chain, channel count): a channel set that does not
    change costs no set-up after the first chunk, and a RETUNE (offset_hz / squelch_db changed by an API PATCH,
    capture.py:442-501) costs one 4 K-byte copy (`ChannelBank.set_offsets`), not a new bank.
    `process` holds the dispatcher's lock for the whole chunk: the seam submits chunks to a multi-worker executor and
    cannot cancel a job that is already running after a timeout, so two chunks may be in flight at once; the banks
    own device scratch and queue onto one stream, so they run one chunk at a time, in submission order of the lock."""

    def __init__(self, sample_rate: int, apply_squelch: bool = False, max_banks: int = 32):
        self._torch = _lib.require_gpu()
        self.sample_rate = int(sample_rate)
        self.apply_squelch = bool(apply_squelch)
        self.max_banks = int(max_banks)
        self._banks: dict[tuple, ChannelBank] = {}
        self._lock = threading.Lock()
        self.banks_created = 0

    def process(self, samples, cfgs, snr: bool = False) -> list[tuple[np.ndarray | None, dict[str, Any]]]:
        """snr=True adds "snr_db" (Channel.update_signal_metrics, capture.py:776-796) to every channel's metrics from
        ONE ranking of the uploaded chunk's magnitudes (no second mix pass, see update_signal_metrics)."""
        cfgs = list(cfgs)
        if not cfgs:
            return []
        x = np.ascontiguousarray(samples, dtype=np.complex64)
        n = x.shape[0]
        if n == 0:
            return [(None, {}) for _ in cfgs]
        if not np.isfinite(x.view(np.float32)).all():          # capture.py:323-325, once for all channels
            logger.warning("ChannelDispatcher: non-finite IQ samples, dropping DSP chunk")
            return [(None, {}) for _ in cfgs]
        with self._lock:
            return self._process_locked(x, n, cfgs, snr)

    def _process_locked(self, x, n, cfgs, snr):
        torch = self._torch
        results: list[Any] = [None] * len(cfgs)
        groups: dict[tuple, list[int]] = {}
        for i, c in enumerate(cfgs):
            if c.mode == "raw" or c.mode in DIGITAL_MODES:
                results[i] = _raw_or_digital(x, self.sample_rate, c)
                continue
            why = _unsupported(c)
            if why:
                raise NotImplementedError(f"wavehip: {why} is not implemented on the device")
            groups.setdefault(_chain_key(c), []).append(i)
        if groups:
            d_in = torch.from_numpy(x).cuda()
            launched = self._launch_groups(d_in, n, cfgs, groups)        # all groups queued before any read-back
            snr_db = update_signal_metrics(d_in, self.sample_rate, [0.0])[0]["snr_db"] if snr else None
            for bank, idx, (audio, met) in launched:
                for i, r in zip(idx, bank.collect(audio, met)):
                    if snr:
                        r[1]["snr_db"] = snr_db
                    results[i] = r
        return results

    def _launch_groups(self, d_in, n, cfgs, groups):
        launched = []
        for key, idx in groups.items():
            group = [cfgs[i] for i in idx]
            offs = tuple(int(v) for v in _offsets_i32(group))
            sq = _squelch_key(group) if self.apply_squelch else None
            has_sq = sq is not None and any(v is not None for v in sq)
            bkey = (n, key, len(idx), has_sq)
            bank = self._banks.get(bkey)
            if bank is None:
                if len(self._banks) >= self.max_banks:
                    self._banks.pop(next(iter(self._banks)))
                bank = self._banks[bkey] = ChannelBank(self.sample_rate, n, group, apply_squelch=self.apply_squelch)
                self.banks_created += 1
            else:
                if bank.offsets != offs:            # retune: one small copy, nothing rebuilt
                    bank.set_offsets(offs)
                if has_sq and bank.squelch != sq:
                    bank.set_squelch(sq)
            launched.append((bank, idx, bank.process_device(d_in, 1)))
        return launched

    ROW_EXTRA = 5       # process_device rows: [audio (row_len) | rssi_db, signal_power_db, max |audio|, finite | n_out]

    def process_device(self, d_in, cfgs, row_len: int | None = None):
        """Device-resident form for callers that keep results on the GPU (the channel split's gather, wire packing):
        `d_in` complex64 GPU tensor of one chunk -> ONE float32 GPU tensor [len(cfgs), row_len + 5], row i = channel
        i's audio (zero padded to row_len = the longest row unless given), then the finalize kernel's four metrics
        (rssi_db, signal_power_db, max |audio|, all-finite flag) and the row's audio length.  Nothing is downloaded, no
        validation is applied (`rows_to_results` does both on whichever rank consumes the rows).  Analog chains only
        ("raw" / digital-voice channels produce host results: use process())."""
        torch = self._torch
        cfgs = list(cfgs)
        assert d_in.is_cuda and d_in.dtype == torch.complex64 and d_in.dim() == 1 and d_in.is_contiguous()
        n = d_in.shape[0]
        groups: dict[tuple, list[int]] = {}
        for i, c in enumerate(cfgs):
            why = "mode 'raw' / digital voice in the device-resident form" if (c.mode == "raw" or c.mode in DIGITAL_MODES) else _unsupported(c)
            if why:
                raise NotImplementedError(f"wavehip: {why} is not implemented on the device")
            groups.setdefault(_chain_key(c), []).append(i)
        with self._lock:
            launched = self._launch_groups(d_in, n, cfgs, groups)
            need = max((b.n_out for b, _, _ in launched), default=0)
            L = need if row_len is None else int(row_len)
            if L < need:
                raise ValueError(f"ChannelDispatcher.process_device: row_len {L} < longest audio row {need}")
            rows = torch.zeros((len(cfgs), L + self.ROW_EXTRA), dtype=torch.float32, device=d_in.device)
            for bank, idx, (audio, met) in launched:
                ii = torch.as_tensor(idx, device=d_in.device)
                rows[ii, :bank.n_out] = audio[0]
                rows[ii, L:L + 4] = met[0]
                rows[ii, L + 4] = float(bank.n_out)
        return rows

    @classmethod
    def rows_to_results(cls, rows) -> list[tuple[np.ndarray | None, dict[str, Any]]]:
        """process_device rows (any device) -> the reference's [(audio | None, metrics)] (validation.py:41-52 applied)."""
        r = rows.cpu().numpy()
        L = r.shape[1] - cls.ROW_EXTRA
        out = []
        for k in range(r.shape[0]):
            m: dict[str, Any] = {"rssi_db": float(r[k, L])}
            if r[k, L + 3] < 0.5 or r[k, L + 2] > AUDIO_MAX_ABS:
                out.append((None, m))
                continue
            m["signal_power_db"] = float(r[k, L + 1])
            out.append((r[k, :int(r[k, L + 4])].copy(), m))
        return out


def noise_blanker(x, threshold_db: float = 10.0, blanking_width: int = 3) -> np.ndarray:
    """dsp/filters.py:267-343 for real float32 signals (the form every demodulator calls it in).  The reference's
    dispatcher never forwards `enable_noise_blanker`, so ChannelBank ignores that flag like the reference does; this
    is the standalone function."""
    torch = _lib.require_gpu()
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.size == 0:
        return a
    d = torch.from_numpy(a).cuda()
    out = torch.empty_like(d)
    factor = float(np.float32(10 ** (threshold_db / 20.0)))
    _lib.check(_lib.lib.wh_noise_blanker(d.data_ptr(), out.data_ptr(), a.size, factor, int(blanking_width),
                                         _lib.stream_ptr(torch)), "wh_noise_blanker")
    return out.cpu().numpy()
