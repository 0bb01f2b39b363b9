"""CPU oracle (numpy) for the WaveCap-SDR per-channel DSP hot path.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement of the reference's
algorithm, written from its published behaviour; it is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- never by the product package (``wavecap-sdr_amd/wavehip``), which
fails loudly when the HIP library is missing.

Parity pin: every function below is checked against golden vectors produced by
importing the real reference in the build container (``oracle/gen_golden.py``
-> ``tests/golden/*.npz``, test ``tests/test_oracle_golden.py``).  The reference
itself holds no numeric golden vectors for this path, so those
generated fixtures are the pin.

All ``file:line`` citations are relative to the reference checkout
(``backend/wavecapsdr/...``).  Third-party arithmetic the reference delegates to
(scipy 1.15.3 ``resample_poly``/``upfirdn``/``lfilter``/``firwin``, numpy 2.2.6
``fft``) is restated from the published algorithm; filter *design* helpers
(``firwin``, ``butter``, ``remez``) are called from scipy because the product's
host code calls the very same functions and they are not on the per-sample path.
"""

from __future__ import annotations

from math import gcd

import numpy as np
from scipy import signal as _sig

# --------------------------------------------------------------------------
# A1  int16 IQ wire conventions
# --------------------------------------------------------------------------


def pack_iq16(samples: np.ndarray) -> np.ndarray:
    """capture.py:102-116 -- clip(-1,1) * 32767.0 -> astype(int16) (truncate toward 0).

    Returns the interleaved int16 array (the reference returns its bytes)."""
    f = np.ascontiguousarray(samples.astype(np.complex64)).view(np.float32).copy()
    np.clip(f, -1.0, 1.0, out=f)
    return (f * np.float32(32767.0)).astype(np.int16)


def unpack_iq16(i16: np.ndarray) -> np.ndarray:
    """cli.py:447-452 / harness.py:274 -- int16.astype(f32)/32768.0, I,Q interleaved."""
    f = i16.astype(np.float32) / np.float32(32768.0)
    return (f[0::2] + 1j * f[1::2]).astype(np.complex64)


def pack_pcm16(x: np.ndarray) -> np.ndarray:
    """capture.py:119-131."""
    f = np.ascontiguousarray(x, dtype=np.float32).copy()
    np.clip(f, -1.0, 1.0, out=f)
    f *= np.float32(32767.0)
    return f.astype(np.int16)


# --------------------------------------------------------------------------
# A2  stateless NCO (float32 phase product)
# --------------------------------------------------------------------------


def nco_table(n: int, offset_hz: int, sample_rate: int) -> np.ndarray:
    """capture.py:166-177.  The Python complex scalar is weak, so the product with the
    float32 ramp happens in complex64: phase[n] = f32(c) * f32(n) rounded to float32,
    then cos/sin of that float32 value."""
    c = np.float32(-2.0 * np.pi * (offset_hz / float(sample_rate)))
    ph32 = c * np.arange(n, dtype=np.float32)  # float32 product, one rounding
    ph = ph32.astype(np.float64)
    return (np.cos(ph) + 1j * np.sin(ph)).astype(np.complex64)


def freq_shift(iq: np.ndarray, offset_hz: float, sample_rate: int) -> np.ndarray:
    """capture.py:180-193 (phase restarts at n=0 every call; offset rounded to int)."""
    if offset_hz == 0.0 or iq.size == 0:
        return iq
    ph = nco_table(iq.shape[0], round(offset_hz), sample_rate)
    return (iq.astype(np.complex64, copy=False) * ph).astype(np.complex64)


def freq_shift_am(iq: np.ndarray, offset_hz: float, sample_rate: int) -> np.ndarray:
    """dsp/am.py:23-42 -- float64 phase, *positive* sign (used for the SSB BFO)."""
    if offset_hz == 0.0 or iq.size == 0:
        return iq
    n = np.arange(iq.shape[0], dtype=np.float64)
    ph = np.exp(1j * 2.0 * np.pi * (offset_hz / float(sample_rate)) * n).astype(np.complex64)
    return (iq.astype(np.complex64, copy=False) * ph).astype(np.complex64)


# --------------------------------------------------------------------------
# A3  FM discriminator, RMS normalise, soft clip
# --------------------------------------------------------------------------


def quadrature_demod(iq: np.ndarray, sample_rate: int) -> np.ndarray:
    """dsp/fm.py:65-97 -- out[0]=0, out[n]=atan2f(x[n] conj x[n-1]) * f32(fs/(2 pi 75000))."""
    if iq.size == 0:
        return np.empty(0, dtype=np.float32)
    x = iq.astype(np.complex64, copy=False)
    prod = x[1:] * np.conj(x[:-1])
    out = np.empty(iq.size, dtype=np.float32)
    out[0] = 0.0
    out[1:] = np.arctan2(prod.imag, prod.real) * np.float32(sample_rate / (2.0 * np.pi * 75000.0))
    return out


def rms_normalize(x: np.ndarray, target_rms: float = 0.18, min_rms: float = 1e-4) -> np.ndarray:
    """dsp/fm.py:42-62."""
    if x.size == 0:
        return x
    rms = float(np.sqrt(np.mean(x ** 2)))
    if rms > min_rms:
        return x * (target_rms / rms)
    return x


def soft_clip_fm(x: np.ndarray) -> np.ndarray:
    """dsp/fm.py:26-39 -- tanh(1.5 x)/tanh(1.5) * 0.95 in float32."""
    k = np.float32(1.5)
    norm = np.float32(1.0 / np.tanh(1.5))
    head = np.float32(0.95)
    return np.tanh(x * k) * norm * head


def soft_clip_agc(x: np.ndarray) -> np.ndarray:
    """dsp/agc.py:58-70 -- same knee without the 0.95 headroom."""
    return np.tanh(x * np.float32(1.5)) * np.float32(1.0 / np.tanh(1.5))


# --------------------------------------------------------------------------
# A6  polyphase rational resampler (scipy.signal.resample_poly restated)
# --------------------------------------------------------------------------


def resample_design(in_rate: int, out_rate: int) -> tuple[np.ndarray, int, int, int]:
    """Taps and alignment of ``scipy.signal.resample_poly`` (scipy 1.15.3
    ``_signaltools.resample_poly``): returns (h float64 already scaled by ``up``, up,
    down, d0) such that  y[m] = sum_j h[j] * xup[m*down + d0 - j]  where ``xup`` is the
    zero-stuffed input (xup[i*up] = x[i])."""
    g = gcd(int(in_rate), int(out_rate))
    up = int(out_rate) // g
    down = int(in_rate) // g
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = _sig.firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float64)
    h = h * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    d0 = n_pre_remove * down - n_pre_pad
    return h, up, down, d0


def resample_out_len(n_in: int, up: int, down: int) -> int:
    n_out = n_in * up
    return n_out // down + (1 if n_out % down else 0)


def resample_poly(x: np.ndarray, in_rate: int, out_rate: int) -> np.ndarray:
    """dsp/fm.py:184-221 -> scipy.signal.resample_poly(x.astype(f64), up, down).astype(f32).

    Restated as a direct zero-padded FIR evaluated only at the kept outputs."""
    if x.size == 0 or in_rate == out_rate:
        return x.astype(np.float32, copy=False)
    h, up, down, d0 = resample_design(in_rate, out_rate)
    n_in = x.shape[0]
    n_out = resample_out_len(n_in, up, down)
    nt = h.shape[0]
    xup = np.zeros(n_in * up, dtype=np.float64)
    xup[::up] = x.astype(np.float64)
    # window for output m covers xup[m*down + d0 - (nt-1) .. m*down + d0]
    lo_pad = nt - 1
    hi_pad = max(0, (n_out - 1) * down + d0 + 1 - xup.shape[0])
    xp = np.concatenate([np.zeros(lo_pad), xup, np.zeros(hi_pad)])
    starts = np.arange(n_out) * down + d0  # index of newest sample, in xup coordinates
    win = np.lib.stride_tricks.sliding_window_view(xp, nt)[starts]  # win[m, i] = xup[m*down+d0-(nt-1)+i]
    y = win @ h[::-1]
    return y.astype(np.float32)


# --------------------------------------------------------------------------
# IIR helpers (A14): cached Butterworth / de-emphasis, applied with lfilter
# --------------------------------------------------------------------------


def lfilter_df2t(b: np.ndarray, a: np.ndarray, x: np.ndarray, dtype=np.float64) -> np.ndarray:
    """Direct-form-II-transposed recurrence of ``scipy.signal.lfilter`` (zero initial
    state) in ``dtype`` -- pure-Python loop, for short cross-checks only."""
    b = np.asarray(b, dtype=dtype)
    a = np.asarray(a, dtype=dtype)
    b = b / a[0]
    a = a / a[0]
    n = max(len(a), len(b))
    bb = np.zeros(n, dtype=dtype)
    aa = np.zeros(n, dtype=dtype)
    bb[: len(b)] = b
    aa[: len(a)] = a
    z = np.zeros(n, dtype=dtype)
    y = np.empty(len(x), dtype=dtype)
    for i, xi in enumerate(np.asarray(x, dtype=dtype)):
        yi = z[0] + bb[0] * xi
        for k in range(1, n):
            z[k - 1] = (z[k] if k < n - 1 else dtype(0)) + bb[k] * xi - aa[k] * yi
        y[i] = yi
    return y


def deemphasis_coeffs(sample_rate: int, tau: float) -> tuple[np.ndarray, np.ndarray]:
    """dsp/fm.py:101-108."""
    tau_us = int(tau * 1e6)
    t = tau_us * 1e-6
    alpha = 1.0 / (1.0 + (1.0 / (2.0 * np.pi * t * sample_rate)))
    return np.array([alpha], dtype=np.float32), np.array([1.0, -(1.0 - alpha)], dtype=np.float32)


def deemphasis_filter(x: np.ndarray, sample_rate: int, tau: float = 75e-6) -> np.ndarray:
    """dsp/fm.py:111-126 (float32 one-pole)."""
    b, a = deemphasis_coeffs(sample_rate, tau)
    return _sig.lfilter(b, a, x).astype(np.float32)


def lpf_audio(x: np.ndarray, sample_rate: int, cutoff: float = 15_000) -> np.ndarray:
    """dsp/fm.py:129-181 -- butter(5, cutoff) applied in float64, zero state per chunk."""
    nc = int(cutoff) / (sample_rate / 2.0)
    if nc >= 1.0 or x.size == 0:
        return x.astype(np.float32, copy=False)
    b, a = _sig.butter(5, nc, btype="low")
    return _sig.lfilter(b, a, x).astype(np.float32)


def butter_filter(x: np.ndarray, sample_rate: int, kind: str, cutoff, order: int = 5) -> np.ndarray:
    """dsp/filters.py:86-221 -- highpass/lowpass/bandpass Butterworth order 5 via lfilter."""
    nyq = sample_rate / 2.0
    if kind == "band":
        lo, hi = cutoff
        wn = [lo / nyq, hi / nyq]
        if wn[0] <= 0 or wn[1] >= 1.0:
            return x.astype(np.float32, copy=False)
        b, a = _sig.butter(order, wn, btype="band")
    else:
        wn = cutoff / nyq
        if wn >= 1.0 or wn <= 0:
            return x.astype(np.float32, copy=False)
        b, a = _sig.butter(order, wn, btype=kind)
    return _sig.lfilter(b, a, x).astype(np.float32)


# --------------------------------------------------------------------------
# Chains on the dispatcher (capture.py:298-439 with mode defaults capture.py:3425-3496)
# --------------------------------------------------------------------------


def nbfm_demod(iq: np.ndarray, sample_rate: int, audio_rate: int = 48_000) -> np.ndarray:
    """dsp/fm.py:317-406 with every optional filter off (the NBFM mode default)."""
    fm = quadrature_demod(iq, sample_rate)
    fm = rms_normalize(fm, target_rms=0.18)
    audio = resample_poly(fm, sample_rate, audio_rate)
    return soft_clip_fm(audio)


def wbfm_demod(iq: np.ndarray, sample_rate: int, audio_rate: int = 48_000,
               deemphasis_tau: float = 75e-6, mpx_cutoff_hz: float = 15_000) -> np.ndarray:
    """dsp/fm.py:228-314 with the WBFM mode defaults (de-emphasis 75 us + MPX 15 kHz)."""
    fm = quadrature_demod(iq, sample_rate)
    fm = deemphasis_filter(fm, sample_rate, tau=deemphasis_tau)
    fm = lpf_audio(fm, sample_rate, cutoff=mpx_cutoff_hz)
    fm = rms_normalize(fm, target_rms=0.18)
    audio = resample_poly(fm, sample_rate, audio_rate)
    return soft_clip_fm(audio)


def channel_metrics_db(x: np.ndarray) -> float:
    """capture.py:331-334 / 436-437 -- 10 log10(mean(|x|^2) + 1e-10)."""
    if np.iscomplexobj(x):
        p = np.mean(np.abs(x) ** 2)
    else:
        p = np.mean(x ** 2)
    return float(10.0 * np.log10(p + 1e-10))


def process_channel_nbfm(samples: np.ndarray, sample_rate: int, offset_hz: float,
                         audio_rate: int = 48_000):
    """capture.py:298-439 for mode 'nbfm' -> (audio, {'rssi_db', 'signal_power_db'})."""
    metrics = {}
    if samples.size == 0:
        return None, metrics
    if not np.isfinite(samples).all():
        return None, metrics
    base = samples if offset_hz == 0.0 else freq_shift(samples, offset_hz, sample_rate)
    metrics["rssi_db"] = channel_metrics_db(base)
    audio = nbfm_demod(base, sample_rate, audio_rate)
    if not np.isfinite(audio).all() or float(np.max(np.abs(audio))) > 1.2:
        return None, metrics
    metrics["signal_power_db"] = channel_metrics_db(audio)
    return audio, metrics


def process_channel_wbfm(samples: np.ndarray, sample_rate: int, offset_hz: float,
                         audio_rate: int = 48_000):
    metrics = {}
    if samples.size == 0 or not np.isfinite(samples).all():
        return None, metrics
    base = samples if offset_hz == 0.0 else freq_shift(samples, offset_hz, sample_rate)
    metrics["rssi_db"] = channel_metrics_db(base)
    audio = wbfm_demod(base, sample_rate, audio_rate)
    if not np.isfinite(audio).all() or float(np.max(np.abs(audio))) > 1.2:
        return None, metrics
    metrics["signal_power_db"] = channel_metrics_db(audio)
    return audio, metrics


# --------------------------------------------------------------------------
# A14  AM / SSB / AGC / optional FM filters (dispatcher modes am, ssb; capture.py:371-414)
# --------------------------------------------------------------------------


def notch_filter(x: np.ndarray, sample_rate: int, freq: float, q: float = 30.0) -> np.ndarray:
    """dsp/filters.py:224-264 -- iirnotch applied with lfilter (float64), float32 out."""
    nf = freq / (sample_rate / 2.0)
    if x.size == 0 or nf <= 0 or nf >= 1.0:
        return x.astype(np.float32, copy=False)
    b, a = _sig.iirnotch(nf, q)
    return _sig.lfilter(b, a, x).astype(np.float32)


def agc_coeffs(sample_rate: int, attack_ms: float = 5.0, release_ms: float = 50.0):
    """dsp/agc.py:215-219."""
    attack_samples = (attack_ms / 1000.0) * sample_rate
    release_samples = (release_ms / 1000.0) * sample_rate
    ac = 1.0 - np.exp(-1.0 / attack_samples) if attack_samples > 0 else 1.0
    rc = 1.0 - np.exp(-1.0 / release_samples) if release_samples > 0 else 1.0
    return ac, rc


def apply_agc(x: np.ndarray, sample_rate: int, target_db: float = -20.0, attack_ms: float = 5.0,
              release_ms: float = 50.0, max_gain_db: float = 60.0) -> np.ndarray:
    """dsp/agc.py:169-242 with the scipy envelope detector (:73-108): two cascaded float32
    one-poles on |x|, envelope = max of both, gain = target/max(env, 1e-6) capped, soft clip."""
    if x.size == 0:
        return x.astype(np.float32, copy=False)
    target_linear = 10.0 ** (target_db / 20.0)
    max_gain_linear = 10.0 ** (max_gain_db / 20.0)
    ac, rc = agc_coeffs(sample_rate, attack_ms, release_ms)
    abs_x = np.abs(x).astype(np.float32)
    env_a = _sig.lfilter(np.array([ac], dtype=np.float32), np.array([1.0, -(1.0 - ac)], dtype=np.float32), abs_x)
    env_r = _sig.lfilter(np.array([rc], dtype=np.float32), np.array([1.0, -(1.0 - rc)], dtype=np.float32), env_a)
    envelope = np.maximum(env_a, env_r).astype(np.float32)
    gain = target_linear / np.maximum(envelope, 1e-6)
    np.minimum(gain, max_gain_linear, out=gain)
    y = x * gain
    return soft_clip_agc(y).astype(np.float32)


def am_demod(iq: np.ndarray, sample_rate: int, audio_rate: int = 48_000, enable_agc: bool = True,
             enable_highpass: bool = True, highpass_hz: float = 100, enable_lowpass: bool = True,
             lowpass_hz: float = 5000, agc_target_db: float = -20.0, notch_frequencies=None) -> np.ndarray:
    """dsp/am.py:45-141."""
    if iq.size == 0:
        return np.empty(0, dtype=np.float32)
    audio = np.abs(iq).astype(np.float32)
    if enable_highpass and highpass_hz > 0:
        audio = butter_filter(audio, sample_rate, "high", highpass_hz)
    if enable_lowpass and lowpass_hz > 0:
        audio = butter_filter(audio, sample_rate, "low", lowpass_hz)
    for f in notch_frequencies or []:
        if 0 < f < sample_rate / 2:
            audio = notch_filter(audio, sample_rate, f)
    if enable_agc:
        audio = apply_agc(audio, sample_rate, target_db=agc_target_db)
    audio = resample_poly(audio, sample_rate, audio_rate)
    if not enable_agc:
        audio = soft_clip_agc(audio)
    return audio


def ssb_demod(iq: np.ndarray, sample_rate: int, audio_rate: int = 48_000, mode: str = "usb",
              enable_agc: bool = True, enable_bandpass: bool = True, bandpass_low: float = 300,
              bandpass_high: float = 3000, agc_target_db: float = -20.0, notch_frequencies=None,
              bfo_offset_hz: float = 1500.0) -> np.ndarray:
    """dsp/am.py:144-247 (BFO: float64 phase, + sign, dsp/am.py:23-42)."""
    if iq.size == 0:
        return np.empty(0, dtype=np.float32)
    shift_hz = bfo_offset_hz if mode.lower() == "usb" else -bfo_offset_hz
    t = np.arange(iq.shape[0], dtype=np.float64) / float(sample_rate)
    shift = np.exp(2j * np.pi * shift_hz * t).astype(np.complex64)
    audio = np.real((iq * shift).astype(np.complex64)).astype(np.float32)
    if enable_bandpass:
        audio = butter_filter(audio, sample_rate, "band", (bandpass_low, bandpass_high))
    for f in notch_frequencies or []:
        if 0 < f < sample_rate / 2:
            audio = notch_filter(audio, sample_rate, f)
    if enable_agc:
        audio = apply_agc(audio, sample_rate, target_db=agc_target_db)
    audio = resample_poly(audio, sample_rate, audio_rate)
    if not enable_agc:
        audio = soft_clip_agc(audio)
    return audio


def sam_demod(iq: np.ndarray, sample_rate: int, audio_rate: int = 48_000, sideband: str = "dsb",
              pll_bandwidth: float = 50.0, enable_agc: bool = True, enable_highpass: bool = True,
              highpass_hz: float = 100.0, enable_lowpass: bool = True, lowpass_hz: float = 5000.0,
              agc_target_db: float = -20.0) -> np.ndarray:
    """dsp/sam.py:132-269 as reached through sam_demod_simple (fresh PLL per call, no notch / blanker).
    PLL (:73-122): float64 phase / integrator, LO exp(-1j phase) in complex128, outputs stored as float32."""
    fs = float(sample_rate)
    omega_n = 2 * np.pi * pll_bandwidth
    alpha, beta = 2 * 0.707 * omega_n / fs, (omega_n ** 2) / (fs ** 2)
    n = iq.shape[0]
    ci = np.zeros(n, dtype=np.float32)
    cq = np.zeros(n, dtype=np.float32)
    x = iq.astype(np.complex128)
    phase = integ = 0.0
    import math
    for i in range(n):
        lr, li = math.cos(phase), -math.sin(phase)
        xr, xi = x[i].real, x[i].imag
        mr, mi = xr * lr - xi * li, xr * li + xi * lr
        ci[i], cq[i] = mr, mi
        pe = math.atan2(mi, abs(mr) + 1e-10)
        integ += beta * pe
        phase += alpha * pe + integ
        if phase > math.pi:
            phase -= 2 * math.pi
        elif phase < -math.pi:
            phase += 2 * math.pi
    sb = sideband.lower()
    audio = ci + cq if sb == "usb" else (ci - cq if sb == "lsb" else ci)
    if enable_highpass and highpass_hz > 0:
        audio = butter_filter(audio, sample_rate, "high", highpass_hz)
    if enable_lowpass and lowpass_hz > 0:
        audio = butter_filter(audio, sample_rate, "low", lowpass_hz)
    if enable_agc:
        audio = apply_agc(audio, sample_rate, target_db=agc_target_db)
    audio = resample_poly(audio, sample_rate, audio_rate)
    if not enable_agc:
        audio = soft_clip_agc(audio)
    return audio


def spectral_noise_reduction(x: np.ndarray, reduction_db: float = 12.0, fft_size: int = 1024) -> np.ndarray:
    """dsp/filters.py:346-460 (overlap 0.5): float32 STFT / ISTFT (numpy 2 keeps float32 in rfft/irfft),
    per-bin 10th-percentile noise floor, gain max(1 - (floor*10^(dB/20)/|X|)^2, 0.1), overlap-add over the
    summed squared window.  Returns (n_frames-1)*hop + fft_size samples (<= len(x))."""
    if x.size == 0 or x.size < fft_size:
        return x.astype(np.float32, copy=False)
    hop = int(fft_size * 0.5)
    win = _sig.windows.hann(fft_size, sym=False).astype(np.float32)
    nf = (len(x) - fft_size) // hop + 1
    length = (nf - 1) * hop + fft_size
    idx = np.arange(fft_size)[None, :] + hop * np.arange(nf)[:, None]
    spec = np.fft.rfft(x[idx] * win, axis=1).astype(np.complex64)
    mag = np.abs(spec)
    floor = np.percentile(mag, 10, axis=0)
    scaled = floor * 10 ** (reduction_db / 20.0)
    gain = np.maximum(np.maximum(0.0, 1.0 - (scaled / np.maximum(mag, 1e-10)) ** 2), 0.1)
    clean = (mag * gain) * np.exp(1j * np.angle(spec))
    frames = np.fft.irfft(clean, n=fft_size, axis=1).astype(np.float32) * win
    out = np.zeros(length, dtype=np.float32)
    wsum = np.zeros(length, dtype=np.float32)
    for i in range(nf):
        out[i * hop:i * hop + fft_size] += frames[i]
        wsum[i * hop:i * hop + fft_size] += win ** 2
    out /= np.maximum(wsum, 1e-10)
    return out[:len(x)].astype(np.float32)


def fm_demod_filtered(iq: np.ndarray, sample_rate: int, audio_rate: int = 48_000, *, deemphasis_tau=None,
                      mpx_cutoff_hz=None, highpass_hz=None, lowpass_hz=None, notch_frequencies=None,
                      noise_reduction_db=None) -> np.ndarray:
    """dsp/fm.py:228-406 with any subset of the optional IIR stages, in the reference's order:
    de-emphasis, [wbfm: MPX low-pass], high-pass, [nbfm: low-pass], notches."""
    fm = quadrature_demod(iq, sample_rate)
    if deemphasis_tau is not None:
        fm = deemphasis_filter(fm, sample_rate, tau=deemphasis_tau)
    if mpx_cutoff_hz is not None:
        fm = lpf_audio(fm, sample_rate, cutoff=mpx_cutoff_hz)
    if highpass_hz is not None and highpass_hz > 0:
        fm = butter_filter(fm, sample_rate, "high", highpass_hz)
    if lowpass_hz is not None and lowpass_hz > 0:
        fm = butter_filter(fm, sample_rate, "low", lowpass_hz)
    for f in notch_frequencies or []:
        if 0 < f < sample_rate / 2:
            fm = notch_filter(fm, sample_rate, f)
    if noise_reduction_db is not None:
        fm = spectral_noise_reduction(fm, reduction_db=noise_reduction_db)
    fm = rms_normalize(fm, target_rms=0.18)
    audio = resample_poly(fm, sample_rate, audio_rate)
    return soft_clip_fm(audio)


def process_channel(samples: np.ndarray, sample_rate: int, offset_hz: float, demod, **kw):
    """capture.py:298-439 around an arbitrary demodulator `demod(base, sample_rate, **kw)`."""
    metrics = {}
    if samples.size == 0 or not np.isfinite(samples).all():
        return None, metrics
    base = samples if offset_hz == 0.0 else freq_shift(samples, offset_hz, sample_rate)
    metrics["rssi_db"] = channel_metrics_db(base)
    audio = demod(base, sample_rate, **kw)
    if not np.isfinite(audio).all() or float(np.max(np.abs(audio))) > 1.2:
        return None, metrics
    metrics["signal_power_db"] = channel_metrics_db(audio)
    return audio, metrics


def process_channel_raw_or_digital(samples: np.ndarray, sample_rate: int, offset_hz: float, mode: str):
    """capture.py:415-430: "raw" = shifted IQ interleaved as float32 (validated like audio); the digital voice
    modes return no audio and signal_power_db = power of the shifted IQ."""
    metrics = {}
    if samples.size == 0 or not np.isfinite(samples).all():
        return None, metrics
    base = samples if offset_hz == 0.0 else freq_shift(samples, offset_hz, sample_rate)
    metrics["rssi_db"] = float(10.0 * np.log10(np.mean(np.abs(base) ** 2) + 1e-10))
    if mode != "raw":
        metrics["signal_power_db"] = metrics["rssi_db"]
        return None, metrics
    audio = np.empty(base.size * 2, dtype=np.float32)
    audio[0::2] = base.real
    audio[1::2] = base.imag
    if not np.isfinite(audio).all() or float(np.max(np.abs(audio))) > 1.2:
        return None, metrics
    metrics["signal_power_db"] = float(10.0 * np.log10(np.mean(audio ** 2) + 1e-10))
    return audio, metrics


def update_signal_metrics(iq: np.ndarray, sample_rate: int, offset_hz: float):
    """Channel.update_signal_metrics, capture.py:749-798 (the every-10th-call SNR branch taken)."""
    shifted = iq if offset_hz == 0.0 else freq_shift(iq, offset_hz, sample_rate)
    mag = np.abs(shifted)
    rssi = float(10.0 * np.log10(np.mean(mag ** 2) + 1e-10))
    n = mag.size
    k_noise, k_signal = n // 10, n - n // 10 - 1
    snr = None
    if k_noise > 0 and k_signal > k_noise:
        part = np.partition(mag, [k_noise, k_signal])
        noise_power, signal_power = part[k_noise] ** 2, part[k_signal] ** 2
        if noise_power > 1e-10:
            snr = float(10.0 * np.log10(signal_power / noise_power))
    return rssi, snr


# --------------------------------------------------------------------------
# A7  2x-oversampled polyphase filterbank
# --------------------------------------------------------------------------


class PolyphaseChannelizer:
    """dsp/channelizer.py:28-158, vectorised over hops.

    Quirks kept: hop M/2 but the whole M-block is inserted into column 0
    of the arm history; no per-hop circular shift; taps not reversed per arm; forward
    unnormalised FFT; trailing <M samples dropped; only ``arm_history`` carries over."""

    def __init__(self, sample_rate: float, channel_bandwidth: int = 25000, taps_per_channel: int = 9):
        self.sample_rate = sample_rate
        self.channel_bandwidth = channel_bandwidth
        self.taps_per_channel = taps_per_channel
        self.channel_count = int(sample_rate / channel_bandwidth)
        if self.channel_count % 2 != 0:
            self.channel_count -= 1
        self.channel_sample_rate = (sample_rate / self.channel_count) * 2
        self.arms = design_pfb_arms(sample_rate, channel_bandwidth, taps_per_channel, self.channel_count)
        self.arm_history = np.zeros((self.channel_count, taps_per_channel), dtype=np.complex64)

    def reset(self) -> None:
        self.arm_history.fill(0)

    def process(self, samples: np.ndarray) -> np.ndarray:
        """Returns c64[hops, M] (row h == the reference's results[h])."""
        M, T = self.channel_count, self.taps_per_channel
        hop = M // 2
        n = len(samples)
        H = 0 if n < M else (n - M) // hop + 1
        if H == 0:
            return np.zeros((0, M), dtype=np.complex64)
        x = np.ascontiguousarray(samples, dtype=np.complex64)
        blocks = np.lib.stride_tricks.as_strided(x, shape=(H, M), strides=(hop * x.itemsize, x.itemsize))
        # B[r] = block_{r-(T-1)}; rows 0..T-2 come from the carried history (column j = block_{-1-j}),
        # column T-1 of the history is about to be rolled out and never used.
        B = np.empty((H + T - 1, M), dtype=np.complex64)
        for j in range(T - 1):
            B[T - 2 - j] = self.arm_history[:, j]
        B[T - 1:] = blocks
        acc = np.zeros((H, M), dtype=np.complex128)
        for j in range(T):
            acc += B[T - 1 - j: T - 1 - j + H].astype(np.complex128) * self.arms[:, j][None, :]
        y = acc.astype(np.complex64)
        out = np.fft.fft(y, axis=1).astype(np.complex64)
        # new history: column j = block_{H-1-j}
        for j in range(T):
            self.arm_history[:, j] = B[H + T - 2 - j]
        return out

    @staticmethod
    def extract_channel(results: np.ndarray, idx: int) -> np.ndarray:
        return np.asarray(results)[:, idx].astype(np.complex64)


def design_pfb_arms(sample_rate: float, channel_bandwidth: int, taps_per_channel: int, M: int) -> np.ndarray:
    """dsp/channelizer.py:69-89."""
    L = M * taps_per_channel - 1
    cutoff = (channel_bandwidth * 0.9) / (sample_rate / 2)
    proto = _sig.firwin(L, cutoff, window=("kaiser", 8.0)).astype(np.float64)
    arms = np.zeros((M, taps_per_channel), dtype=np.float64)
    for k in range(M):
        t = proto[k::M]
        arms[k, : len(t)] = t
    return arms


# --------------------------------------------------------------------------
# A8  spectrum
# --------------------------------------------------------------------------


def spectrum(iq: np.ndarray, sample_rate: int, fft_size: int):
    """dsp/fft/scipy_backend.py:38-79 -> (power_db f32[N], freqs f32[N], bin_hz)."""
    N = fft_size
    if iq.size < N:
        return np.zeros(N, np.float32), np.zeros(N, np.float32), sample_rate / N
    from scipy import fft as _sfft  # the reference's backend is scipy.fft (pocketfft C++)

    w = np.hanning(N).astype(np.float32)
    X = _sfft.fftshift(_sfft.fft(iq[:N] * w))
    p = 20.0 * np.log10(np.abs(X) + 1e-10)
    freqs = _sfft.fftshift(_sfft.fftfreq(N, 1.0 / sample_rate))
    return p.astype(np.float32), freqs.astype(np.float32), sample_rate / N


# --------------------------------------------------------------------------
# A13  activity statistics (per-channel power of a PFB output block)
# --------------------------------------------------------------------------


def pfb_channel_stats(out: np.ndarray) -> np.ndarray:
    """Per-channel {sum p, sum p^2, count, min p, max p} of p=|y|^2 over hops, the
    BinStats fields of channel_classifier.py:17-48 applied to filterbank channels.
    Returns float64[M, 5].  Definition (round 3): p is taken in FLOAT32, p = f32(f32(re^2) + f32(im^2)) -- min / max are
    therefore exact float32 values -- and the sums are sums of those float32 p (here in float64; the device adds short
    float32 blocks in float64 and agrees to <= 2e-6 relative)."""
    re, im = out.real.astype(np.float32), out.imag.astype(np.float32)
    p = (re * re + im * im).astype(np.float64)          # float32 products, float32 sum (numpy keeps float32 here)
    H = p.shape[0]
    return np.stack([p.sum(0), (p * p).sum(0), np.full(p.shape[1], float(H)), p.min(0), p.max(0)], axis=1)


# --------------------------------------------------------------------------
# A4 / A5  trunking front-end: phase-continuous NCO + two-stage FIR decimation
# --------------------------------------------------------------------------


def recorder_decimation_plan(sample_rate: int, target_rate: int = 48000):
    """VoiceRecorder.setup_decimation_filter stage factors (trunking/system.py:453-485)."""
    total = max(1, sample_rate // target_rate)
    if total <= 1:
        return 1, 1
    if total >= 100:
        for s1 in [25, 20, 30, 16]:
            if total % s1 == 0 and total // s1 <= 10:
                return s1, total // s1
    return total, 1


class TrunkingDDC:
    """trunking/system.py:1392-1466, 1735-1779 with dsp/filters.py:558-646 (scipy fallback path):
    float64-phase NCO continued across calls (index wrapped at one second), each stage
    lfilter(float64 taps) on complex128 with carried state (first call: lfilter_zi * x[0]),
    rounded to complex64, then [::D] restarting at index 0 every call."""

    def __init__(self, sample_rate: int, stage1_factor: int, stage2_factor: int = 1):
        self.fs = int(sample_rate)
        self.d1, self.d2 = int(stage1_factor), int(stage2_factor)
        self.t1 = _sig.firwin(157, 0.8 / self.d1, window=("kaiser", 7.857))
        self.t2 = _sig.firwin(73, 0.8 / self.d2, window=("kaiser", 7.857)) if self.d2 > 1 else None
        self.reset()

    def reset(self):
        self.h1 = None      # last len(t1)-1 mixed inputs (equivalent to the carried zi)
        self.h2 = None
        self.idx = 0
        self.last_off = 0.0

    @staticmethod
    def _stage(x, taps, hist, D):
        L = len(taps)
        if hist is None:
            hist = np.full(L - 1, x[0], dtype=np.complex128)     # lfilter_zi(taps) * x[0]
        ext = np.concatenate([hist, x.astype(np.complex128)])
        y = np.convolve(ext, taps.astype(np.float64))[L - 1: L - 1 + len(x)]
        return y.astype(np.complex64)[::D], ext[len(ext) - (L - 1):]

    def process(self, iq: np.ndarray, offset_hz: float) -> np.ndarray:
        if iq.size == 0:
            return iq
        if offset_hz == 0.0:
            c = iq
        else:
            if offset_hz != self.last_off:
                self.idx = 0
                self.last_off = offset_hz
            n = np.arange(iq.size, dtype=np.float64) + self.idx
            phase = -2.0 * np.pi * offset_hz * n / self.fs
            c = (iq.astype(np.complex64, copy=False) * np.exp(1j * phase).astype(np.complex64)).astype(np.complex64)
            self.idx += iq.size
            if self.idx >= self.fs:
                self.idx %= self.fs
        y, self.h1 = self._stage(c, self.t1, self.h1, self.d1)
        if self.t2 is not None and y.size > 0:
            y, self.h2 = self._stage(y, self.t2, self.h2, self.d2)
        return y


# --------------------------------------------------------------------------
# A13  control-channel scanner measurement (trunking/cc_scanner.py:165-264)
# --------------------------------------------------------------------------


def scanner_power(iq: np.ndarray, sample_rate: int, offsets_hz) -> np.ndarray:
    """Per offset: freq_shift -> lfilter(firwin(65, 0.8/D, kaiser 6.0)) (zero state, complex128)
    -> [::D] -> (mean |y|^2, max |y|^2)."""
    D = max(1, sample_rate // 48000)
    taps = _sig.firwin(65, 0.8 / D, window=("kaiser", 6.0)) if D > 1 else None
    out = np.zeros((len(offsets_hz), 2))
    for i, off in enumerate(offsets_hz):
        s = freq_shift(iq, float(off), sample_rate)
        if taps is not None:
            y = np.convolve(s.astype(np.complex128), taps)[: len(s)][::D]
        else:
            y = s
        p = np.abs(y) ** 2
        out[i] = (np.mean(p), np.max(p))
    return out


def scanner_sync_correlation(y: np.ndarray) -> float:
    """cc_scanner.py:266-353 on one decimated (complex128) stream: best normalised correlation of the
    FM-demodulated symbol samples (10 samples/symbol, taken at 5::10) with the +-0.2356 sync waveform."""
    sync = np.array([1, 1, 1, 1, 1, 3, 1, 1, 3, 3, 1, 1, 3, 3, 3, 3, 1, 3, 1, 3, 3, 3, 3, 3])
    if len(y) < 10 * 24 + 10:
        return 0.0
    fm = np.angle(y[1:] * np.conj(y[:-1]))
    symbols_count = len(fm) // 10
    if symbols_count < 24:
        return 0.0
    ss = fm[5::10][:symbols_count]
    wave = np.where(sync == 1, 0.2356, -0.2356)
    search_len = min(len(ss) - 24, symbols_count - 24)
    best = 0.0
    norm_sync = np.sqrt(np.sum(wave ** 2))
    for i in range(max(0, search_len)):
        w = ss[i:i + 24]
        c = np.sum(w * wave) / (np.sqrt(np.sum(w ** 2) + 1e-10) * norm_sync)
        if abs(c) > abs(best):
            best = c
    return float(best)


def scanner_measure(iq: np.ndarray, sample_rate: int, channel_offsets_hz, sync_check: bool = False) -> list:
    max_offset = sample_rate / 2 - 15000
    edges = [-max_offset + 25000, max_offset - 25000]
    p = scanner_power(iq, sample_rate, list(channel_offsets_hz) + edges)
    noise = min(p[-2, 0], p[-1, 0])
    eps = 1e-12
    res = []
    for i in range(len(channel_offsets_hz)):
        pw = 10 * np.log10(p[i, 0] + eps)
        nf = 10 * np.log10(noise + eps)
        m = dict(power_db=float(pw), peak_power_db=float(10 * np.log10(p[i, 1] + eps)),
                 noise_floor_db=float(nf), snr_db=float(pw - nf))
        if sync_check:
            D = max(1, sample_rate // 48000)
            taps = _sig.firwin(65, 0.8 / D, window=("kaiser", 6.0))
            sft = freq_shift(iq, float(channel_offsets_hz[i]), sample_rate)
            y = np.convolve(sft.astype(np.complex128), taps)[: len(sft)][::D]
            m["sync_correlation"] = scanner_sync_correlation(y)
            m["sync_detected"] = bool(m["snr_db"] >= 8.0 and abs(m["sync_correlation"]) > 0.6)
        res.append(m)
    return res


# ----------------------------------------------------------------------------------------------
# N2: P25P1SoftSyncDetector (reference decoders/p25_framer.py:124-231); N4: pack_f32 (capture.py:134-144)
# ----------------------------------------------------------------------------------------------

P25_SYNC_SYMBOLS = np.array([3.0 if ((0x5575F5FF77FF >> ((23 - i) * 2)) & 3) == 1 else -3.0 for i in range(24)],
                            dtype=np.float32)


class SoftSyncDetector:
    """State = the last 24 soft symbols (zeros at reset); score i = <pattern, symbols i-23..i>."""

    def __init__(self):
        self.hist = np.zeros(24, dtype=np.float32)

    def reset(self):
        self.hist[:] = 0.0

    def process_batch(self, soft: np.ndarray) -> np.ndarray:
        soft = np.asarray(soft, dtype=np.float32)
        n = soft.size
        if n == 0:
            return np.array([], dtype=np.float32)
        ext = np.concatenate([self.hist, soft])
        scores = np.correlate(ext, P25_SYNC_SYMBOLS, mode="valid")[-n:]
        self.hist = ext[-24:].copy()
        return scores.astype(np.float32)


def pack_f32(samples: np.ndarray) -> bytes:
    if samples.size == 0:
        return b""
    return np.clip(samples.astype(np.float32), -1.0, 1.0).tobytes()


# ----------------------------------------------------------------------------------------------
# N4: ChannelClassifier (reference channel_classifier.py:16-238)
# ----------------------------------------------------------------------------------------------


def binstats_update(stats, frames: np.ndarray) -> np.ndarray:
    """BinStats.update for every bin (channel_classifier.py:26-33): float64 running sum / sum_sq / count /
    min / max, frames folded in order.  stats float64 [N, 5] or None."""
    frames = np.atleast_2d(np.asarray(frames, dtype=np.float32)).astype(np.float64)
    if stats is None:
        stats = np.zeros((frames.shape[1], 5))
        stats[:, 3], stats[:, 4] = np.inf, -np.inf
    for v in frames:
        stats[:, 0] += v
        stats[:, 1] += v * v
        stats[:, 2] += 1
        stats[:, 3] = np.minimum(stats[:, 3], v)
        stats[:, 4] = np.maximum(stats[:, 4], v)
    return stats


def classify_bins(stats: np.ndarray, freqs, center_hz: float, min_samples_per_bin: int = 50,
                  control_variance_threshold: float = 4.0, voice_variance_threshold: float = 10.0) -> list:
    """ChannelClassifier.classify (channel_classifier.py:145-227) -> [(freq_hz, power_db, std_dev_db, type)]."""
    cnt = stats[:, 2]
    mean = np.where(cnt > 0, stats[:, 0] / np.maximum(cnt, 1), 0.0)
    var = np.where(cnt < 2, 0.0, stats[:, 1] / np.maximum(cnt, 1) - mean * mean)
    std = np.sqrt(np.maximum(0.0, var))
    ok = [i for i in range(len(mean)) if cnt[i] >= min_samples_per_bin]
    if not ok:
        return []
    av = sorted(mean[i] for i in ok)
    noise = av[int(len(av) * 0.2)]
    out, visited = [], set()
    for i in sorted(ok, key=lambda j: mean[j], reverse=True):
        if i in visited or mean[i] < noise + 10:
            continue
        prev_avg = mean[i - 1] if i - 1 >= 0 else -np.inf
        next_avg = mean[i + 1] if i + 1 < len(mean) else -np.inf
        if mean[i] <= prev_avg or mean[i] <= next_avg:
            continue
        visited.update(range(i - 3, i + 4))
        kind = ("unknown" if mean[i] < noise + 5 else "control" if std[i] < control_variance_threshold
                else "voice" if std[i] > voice_variance_threshold else "variable")
        out.append((center_hz + (float(freqs[i]) if i < len(freqs) else 0.0), float(mean[i]), float(std[i]), kind))
    out.sort(key=lambda c: c[1], reverse=True)
    return out


# ----------------------------------------------------------------------------------------------
# N2: BCH(63,16,23) NID decode (reference dsp/fec/bch.py) and the framer's NID front half
# (decoders/p25_framer.py:471-617, decoders/nac_tracker.py)
# ----------------------------------------------------------------------------------------------

_BCH_TABLE = None


def bch_table() -> np.ndarray:
    """All 65536 systematic codewords, built from the generator matrix: row i = codeword of the unit word 2^i
    (x^(47+i) plus its remainder modulo g(x) = 0o6331141367235453), the rest by linearity."""
    global _BCH_TABLE
    if _BCH_TABLE is None:
        g = 0o6331141367235453
        rows = []
        for i in range(16):
            r = 1 << (47 + i)
            for bit in range(62, 46, -1):
                if (r >> bit) & 1:
                    r ^= g << (bit - 47)
            rows.append((1 << (47 + i)) | r)
        t = np.zeros(1 << 16, dtype=np.uint64)
        for i, row in enumerate(rows):
            t[1 << i:1 << (i + 1)] = t[:1 << i] ^ np.uint64(row)
        _BCH_TABLE = t
    return _BCH_TABLE


def _popcount64(x: np.ndarray) -> np.ndarray:
    x = x - ((x >> np.uint64(1)) & np.uint64(0x5555555555555555))
    x = (x & np.uint64(0x3333333333333333)) + ((x >> np.uint64(2)) & np.uint64(0x3333333333333333))
    x = (x + (x >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return (x * np.uint64(0x0101010101010101)) >> np.uint64(56)


def bch_decode(word: int, tracked_nac: int = 0):
    """bch.py:533-638 as a bounded-distance decoder (t = 11, d_min = 23): nearest codeword within 11 bits, else the
    same search with the tracked NAC written over the first 12 bits, else (0, -1)."""
    def near(w):
        d = _popcount64(bch_table() ^ np.uint64(w))
        i = int(np.argmin(d))
        return (i, int(d[i])) if d[i] <= 11 else (0, -1)
    word &= (1 << 63) - 1
    dat, e = near(word)
    if e < 0 and tracked_nac and ((word >> 51) & 0xFFF) != tracked_nac:
        dat, e = near((word & ((1 << 51) - 1)) | (tracked_nac << 51))
    return dat, e


def strip_status_symbols(dibits: np.ndarray, initial_counter: int = 21) -> np.ndarray:
    """decoders/p25.py:2796-2862 (_get_status_keep_indices + _strip_status_symbols): the counter loop itself."""
    d = np.asarray(dibits)
    keep = []
    counter = initial_counter
    for i in range(len(d)):
        counter += 1
        if counter == 36:      # status symbol every 36 dibits
            counter = 0
            continue
        keep.append(i)
    if len(d) == 0:
        return np.array([], dtype=np.uint8)
    return np.asarray(d[np.array(keep, dtype=np.int64)], dtype=np.uint8) if keep else np.array([], dtype=np.uint8)


class NIDFrontEnd:
    """process_batch's NID part: sync positions (score > 60) restart a 33-dibit collection that begins AT the dibit
    completing the sync; status dibit 11 dropped; BCH; NAC tracker (3 entries, dominant after 3 observations)."""

    def __init__(self):
        self.sync = SoftSyncDetector()
        self.buf = None
        self.count, self.seen, self.tick = {}, {}, 0
        self.pos = 0

    def tracked(self):
        if not self.count:
            return 0
        nac = max(self.count, key=lambda k: self.count[k])
        return nac if self.count[nac] >= 3 else 0

    def track(self, nac):
        self.tick += 1
        if nac in self.count:
            self.count[nac] += 1
        else:
            self.count[nac] = 1
        self.seen[nac] = self.tick
        if len(self.count) > 3:
            old = min(self.seen, key=lambda k: self.seen[k])
            del self.count[old], self.seen[old]

    def process_batch(self, soft, dibits):
        scores = self.sync.process_batch(soft)
        events = []
        for i, d in enumerate(np.asarray(dibits, dtype=np.uint8)):
            if scores[i] > 60.0:
                self.buf = []
            if self.buf is not None:
                self.buf.append(int(d))
                if len(self.buf) >= 33:
                    nid = self.buf[:11] + self.buf[12:33]
                    word = 0
                    for v in nid:
                        word = (word << 2) | (v & 3)
                    dat, e = bch_decode(word >> 1, self.tracked())
                    self.buf = None
                    if e >= 0:
                        nac = (dat >> 4) & 0xFFF
                        self.track(nac)
                        events.append((self.pos + i, nac, dat & 0xF, e))
        self.pos += len(dibits)
        return events


def noise_blanker(x: np.ndarray, threshold_db: float = 10.0, blanking_width: int = 3) -> np.ndarray:
    """dsp/filters.py:267-343: median |x| baseline, threshold in dB above it, dilated mask zeroed."""
    if x.size == 0:
        return x.astype(np.float32, copy=False)
    mag = np.abs(x)
    med = np.median(mag)
    if med < 1e-10:
        return x.astype(np.float32, copy=False)
    mask = mag > med * (10 ** (threshold_db / 20.0))
    if blanking_width > 0 and mask.any():
        mask = np.convolve(mask.astype(np.float32), np.ones(2 * blanking_width + 1, dtype=np.float32), mode="same") > 0
    y = x.copy()
    y[mask] = 0
    return y.astype(np.float32)


# --------------------------------------------------------------------------
# A12 parts: CostasLoop (dsp/p25/cqpsk.py:84-196) and MuellerMullerTED (dsp/p25/symbol_timing.py:214-380)
# Pure-Python per-sample loops like the reference (small cases only).
# --------------------------------------------------------------------------


class CostasLoop:
    def __init__(self, loop_bw: float = 0.01, damping: float = 0.707, max_freq: float = 0.1):
        theta = loop_bw / (damping + 1 / (4 * damping))          # cqpsk.py:107-110
        d = 1 + 2 * damping * theta + theta ** 2
        self._kp, self._ki, self._max_freq = 4 * damping * theta / d, 4 * theta ** 2 / d, max_freq
        self._phase = self._freq = 0.0

    def process_block(self, samples):
        out = np.zeros(len(samples), dtype=np.complex128)
        q = np.pi / 4
        for i, x in enumerate(np.asarray(samples, dtype=np.complex128)):
            c = x * np.exp(-1j * self._phase)                        # cqpsk.py:138
            ph = np.angle(c)                                         # :166-170
            err = ph - np.round(ph / q) * q
            while err > np.pi: err -= 2 * np.pi
            while err < -np.pi: err += 2 * np.pi
            self._freq = float(np.clip(self._freq + self._ki * err, -self._max_freq, self._max_freq))   # :145-146
            self._phase += self._kp * err + self._freq
            while self._phase > np.pi: self._phase -= 2 * np.pi
            while self._phase < -np.pi: self._phase += 2 * np.pi
            out[i] = c
        return out


class MuellerMullerTED:
    CONST = np.array([1 + 1j, -1 + 1j, -1 - 1j, 1 - 1j], dtype=np.complex128) / np.sqrt(2)

    def __init__(self, samples_per_symbol: float, loop_bw: float = 0.01, damping: float = 1.0):
        theta = loop_bw / (damping + 1 / (4 * damping))          # symbol_timing.py:34-57
        d = 1 + 2 * damping * theta + theta ** 2
        self.sps, self._kp, self._ki = samples_per_symbol, 4 * damping * theta / d, 4 * theta ** 2 / d
        self._phase = self._integ = 0.0
        self._buf = [0j, 0j, 0j, 0j]                              # oldest ... newest
        self._prev_sym = self._prev_dec = 0j

    @staticmethod
    def _interp(v, mu):                                           # symbol_timing.py:296-303
        c0 = v[1]; c1 = (v[2] - v[0]) / 2
        c2 = v[0] - 5 * v[1] / 2 + 2 * v[2] - v[3] / 2
        c3 = (v[3] - v[0]) / 2 + 3 * (v[1] - v[2]) / 2
        return float(c0 + mu * (c1 + mu * (c2 + mu * c3)))

    def process_block(self, samples):
        sym, dec, errs = [], [], []
        for x in np.asarray(samples, dtype=np.complex128):
            self._buf = self._buf[1:] + [complex(x)]
            self._phase += 1.0
            if self._phase >= self.sps:                           # :342-372
                self._phase -= self.sps
                mu = self._phase / self.sps
                cur = complex(self._interp([b.real for b in self._buf], mu), self._interp([b.imag for b in self._buf], mu))
                d = complex(self.CONST[np.argmin(np.abs(self.CONST - cur))])
                e = float(np.real(np.conj(self._prev_dec) * cur - np.conj(d) * self._prev_sym))
                self._integ = float(np.clip(self._integ + self._ki * e, -self.sps / 4, self.sps / 4))
                self._phase += self._kp * e + self._integ
                sym.append(cur); dec.append(d); errs.append(e)
                self._prev_sym, self._prev_dec = cur, d
        return np.array(sym, np.complex128), np.array(dec, np.complex128), np.array(errs, np.float64)
