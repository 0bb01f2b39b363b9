"""Seeded synthetic signal factories shared by the golden generator, the parity
tests and bench.py.

Every factory is a pure function of its arguments (numpy ``default_rng`` with an
explicit seed), so a fixture only has to store the recipe arguments, a sha256 of
the generated input and the expected output.  Recipes follow the reference's
own test factories (reference ``backend/tests/conftest.py:22-88``) and the
configs of BASELINE.json / SURVEY.md §8(d).
"""

from __future__ import annotations

import hashlib

import numpy as np

P25_SYNC_DIBITS = None  # filled below


def sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def noise_c64(n: int, seed: int, amp: float = 0.5) -> np.ndarray:
    """``amp*(randn + j randn)`` as complex64 (benchmark_dsp.py:119 recipe, seeded)."""
    rng = np.random.default_rng(seed)
    re = rng.standard_normal(n, dtype=np.float32)
    im = rng.standard_normal(n, dtype=np.float32)
    return ((re + 1j * im) * np.float32(amp)).astype(np.complex64)


def fm_tone_c64(
    n: int,
    fs: float,
    seed: int,
    audio_hz: float = 1000.0,
    deviation: float = 75_000.0,
    carrier_hz: float = 0.0,
    noise_amp: float = 0.05,
    amp: float = 0.7,
) -> np.ndarray:
    """FM-modulated tone at ``carrier_hz`` plus complex noise (conftest.py:42-56 recipe)."""
    t = np.arange(n, dtype=np.float64) / fs
    phase = (deviation / audio_hz) * np.sin(2 * np.pi * audio_hz * t) + 2 * np.pi * carrier_hz * t
    iq = amp * np.exp(1j * phase)
    rng = np.random.default_rng(seed)
    iq = iq + noise_amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return iq.astype(np.complex64)


def am_tone_c64(n: int, fs: float, seed: int, audio_hz: float = 800.0, depth: float = 0.6,
                carrier_hz: float = 0.0, noise_amp: float = 0.02, amp: float = 0.5) -> np.ndarray:
    t = np.arange(n, dtype=np.float64) / fs
    env = amp * (1.0 + depth * np.sin(2 * np.pi * audio_hz * t))
    iq = env * np.exp(2j * np.pi * carrier_hz * t)
    rng = np.random.default_rng(seed)
    iq = iq + noise_amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return iq.astype(np.complex64)


def nbfm_bank_c64(n: int, fs: float, seed: int, n_ch: int = 32, spacing: float = 50_000.0,
                  deviation: float = 5_000.0, noise_amp: float = 0.01, start: int = 0) -> np.ndarray:
    """Config 2 content: ``n_ch`` NBFM carriers at offsets ``(k-(n_ch-1)/2)*spacing``
    with distinct 300..3000 Hz tones, plus noise."""
    t = (np.arange(n, dtype=np.float64) + start) / fs
    x = np.zeros(n, dtype=np.complex128)
    for k in range(n_ch):
        off = (k - (n_ch - 1) / 2.0) * spacing
        tone = 300.0 + (2700.0 * k) / max(1, n_ch - 1)
        ph = 2 * np.pi * off * t + (deviation / tone) * np.sin(2 * np.pi * tone * t)
        x += (0.6 / n_ch) * np.exp(1j * ph)
    rng = np.random.default_rng(seed)
    x += noise_amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


def nbfm_bank_offsets(n_ch: int = 32, spacing: float = 50_000.0) -> list[float]:
    return [float((k - (n_ch - 1) / 2.0) * spacing) for k in range(n_ch)]


def nbfm_zero_gap_i16(n: int = 120_000, fs: int = 2_400_000, seed: int = 77) -> np.ndarray:
    """Config-2 content as int16 IQ with EXACT zero samples: 600 muted samples and four isolated ones (zero-fill after an
    overflow, muted front ends)."""
    i16 = pack_iq16_np(nbfm_bank_c64(n, fs, seed=seed)).copy()
    i16[2 * 30000:2 * 30600] = 0
    for k in (5, 5000, 70001, n - 1):
        i16[2 * k:2 * k + 2] = 0
    return i16


def pack_iq16_np(x: np.ndarray) -> np.ndarray:
    """Reference wire rule (capture.py:102-116): clip(-1,1)*32767 -> astype(int16)."""
    f = np.ascontiguousarray(x.astype(np.complex64)).view(np.float32).copy()
    np.clip(f, -1.0, 1.0, out=f)
    return (f * np.float32(32767.0)).astype(np.int16)


# --------------------------------------------------------------------------
# P25 C4FM synthetic transmitter
# --------------------------------------------------------------------------

def p25_sync_dibits() -> np.ndarray:
    """24 dibits of the frame sync 0x5575F5FF77FF (c4fm.py:2277-2299)."""
    pattern = 0x5575F5FF77FF
    return np.array([(pattern >> ((23 - i) * 2)) & 3 for i in range(24)], dtype=np.uint8)


_DIBIT_TO_SYMBOL = np.array([1.0, 3.0, -1.0, -3.0])  # dibit 0:+1 1:+3 2:-1 3:-3 (c4fm.py:6-11)


def _rrc_pulse(t: np.ndarray, alpha: float) -> np.ndarray:
    """Root-raised-cosine impulse response at times ``t`` (in symbols)."""
    out = np.empty_like(t)
    eps = 1e-9
    z = np.abs(t) < eps
    s = np.abs(np.abs(t) - 1.0 / (4 * alpha)) < eps
    r = ~(z | s)
    out[z] = 1 - alpha + 4 * alpha / np.pi
    out[s] = (alpha / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * alpha))
                                     + (1 - 2 / np.pi) * np.cos(np.pi / (4 * alpha)))
    tr = t[r]
    out[r] = (np.sin(np.pi * tr * (1 - alpha)) + 4 * alpha * tr * np.cos(np.pi * tr * (1 + alpha))) / (
        np.pi * tr * (1 - (4 * alpha * tr) ** 2))
    return out


def c4fm_frames_dibits(n_symbols: int, seed: int, frame_len: int = 360) -> np.ndarray:
    """Repeating TSDU-like frames: 24-dibit sync + random dibits (config 4)."""
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 4, size=n_symbols, dtype=np.uint8)
    sync = p25_sync_dibits()
    for s in range(0, n_symbols - 24, frame_len):
        d[s:s + 24] = sync
    return d


def c4fm_iq(
    n: int,
    fs: float,
    seed: int,
    snr_db: float = 20.0,
    freq_offset_hz: float = 0.0,
    symbol_rate: float = 4800.0,
    frame_len: int = 360,
    amp: float = 0.5,
    silence: tuple[int, int] | None = None,
    dibits: np.ndarray | None = None,
) -> tuple[np.ndarray, np.ndarray]:
    """RRC-shaped C4FM at complex baseband (deviation: symbol 3 -> 1800 Hz).

    Returns (iq complex64[n], transmitted dibits).  ``silence=(a, b)`` replaces
    samples a..b by noise only (exercises fine-sync loss).  ``dibits`` replaces the
    repeating TSDU-like frames by an explicit dibit sequence (padded with random dibits).
    """
    sps = fs / symbol_rate
    n_sym = int(np.ceil(n / sps)) + 24
    if dibits is None:
        dib = c4fm_frames_dibits(n_sym, seed, frame_len)
    else:
        dib = np.random.default_rng(seed + 31).integers(0, 4, size=n_sym, dtype=np.uint8)
        k = min(n_sym, len(dibits))
        dib[:k] = np.asarray(dibits, dtype=np.uint8)[:k]
    sym = _DIBIT_TO_SYMBOL[dib]
    # frequency waveform f(t) = 600 Hz * sum_k a_k p(t/T - k), RRC pulse alpha 0.2, +-8 symbols
    tn = np.arange(n, dtype=np.float64) / sps
    freq = np.zeros(n, dtype=np.float64)
    k0 = np.floor(tn).astype(np.int64)
    for dk in range(-8, 9):
        k = k0 + dk
        ok = (k >= 0) & (k < n_sym)
        a = np.where(ok, sym[np.clip(k, 0, n_sym - 1)], 0.0)
        freq += a * _rrc_pulse(tn - k, 0.2)
    # normalise pulse so that the matched RRC pair gives unit symbol amplitude
    freq *= 600.0
    phase = 2 * np.pi * np.cumsum(freq + freq_offset_hz) / fs
    iq = amp * np.exp(1j * phase)
    rng = np.random.default_rng(seed + 7919)
    sigma = amp * 10 ** (-snr_db / 20.0) / np.sqrt(2.0)
    noise = sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if silence is not None:
        a, b = silence
        iq[a:b] = 0.0
    return (iq + noise).astype(np.complex64), dib


def c4fm_muted_iq(n: int = 96000, fs: int = 48000, seed: int = 1300):
    """c4fm_iq with EXACT zeros: 6 000 muted samples (far longer than the filters) and three isolated zero samples."""
    iq, dib = c4fm_iq(n, fs, seed, snr_db=20.0, freq_offset_hz=100.0)
    iq = iq.copy()
    iq[30000:36000] = 0
    iq[[7, 50000, 50001]] = 0
    return iq, dib


def p25_head_stream_iq(fs: int = 48000, seed: int = 1500, reps: int = 3, snr_db: float = 20.0,
                       freq_offset_hz: float = 150.0):
    """IQ of a C4FM carrier whose dibits are `reps` different frame-head streams (nid_stream: sync + BCH-coded NID with
    the status dibit, clean / corrupted / restarted heads) behind a 200-dibit preamble of TSDU-like frames for the
    demodulator to lock on -- the input of the chained check IQ -> C4FM demodulator -> soft sync -> NID / BCH."""
    parts = [c4fm_frames_dibits(200, seed)]
    for r in range(reps):
        parts.append(nid_stream(seed + 10 * r)[0])
    dib = np.concatenate(parts)
    n = int(np.ceil((len(dib) + 8) * fs / 4800.0))
    iq, tx = c4fm_iq(n, fs, seed, snr_db=snr_db, freq_offset_hz=freq_offset_hz, dibits=dib)
    return iq, tx


def config4_streams(C: int = 64, fs: int = 48000, n: int = 480000):
    """C independent C4FM streams (seeds 1000+k, frequency offsets U(-400, 400) Hz, SNR 20 dB)."""
    from concurrent.futures import ThreadPoolExecutor

    offs = np.random.default_rng(4).uniform(-400.0, 400.0, size=C)
    with ThreadPoolExecutor(8) as ex:
        rows = list(ex.map(lambda k: c4fm_iq(n, fs, 1000 + k, snr_db=20.0, freq_offset_hz=float(offs[k]))[0], range(C)))
    return np.stack(rows), offs


# --------------------------------------------------------------------------
# P25 Phase-2 style pi/4-DQPSK transmitter (for the CQPSK demodulator rows)
# --------------------------------------------------------------------------

_DQPSK_STEP = np.array([np.pi / 4, 3 * np.pi / 4, -3 * np.pi / 4, -np.pi / 4])  # dibit -> phase change


def dqpsk_iq(n: int, fs: float, seed: int, symbol_rate: float = 12000.0, snr_db: float = 25.0,
             freq_offset_hz: float = 0.0, amp: float = 0.6) -> tuple[np.ndarray, np.ndarray]:
    """pi/4-DQPSK at complex baseband, RRC(alpha=1) shaped, integer samples per symbol."""
    sps = int(round(fs / symbol_rate))
    n_sym = n // sps + 16
    rng = np.random.default_rng(seed)
    dib = rng.integers(0, 4, size=n_sym, dtype=np.uint8)
    ph = np.cumsum(_DQPSK_STEP[dib])
    up = np.zeros(n_sym * sps, dtype=np.complex128)
    up[::sps] = np.exp(1j * ph)
    t = (np.arange(8 * sps + 1) - 4 * sps) / sps
    h = _rrc_pulse(t.astype(np.float64), 1.0)
    h /= np.sum(h)
    x = np.convolve(up, h)[4 * sps: 4 * sps + n] * sps
    tt = np.arange(n) / fs
    x = amp * x * np.exp(2j * np.pi * freq_offset_hz * tt)
    sigma = amp * 10 ** (-snr_db / 20.0) / np.sqrt(2.0)
    x = x + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64), dib


def dqpsk_muted_iq(n: int = 24000, fs: float = 48000, seed: int = 1720):
    """dqpsk_iq with EXACT zeros: a 400-sample muted stretch (longer than the 65-tap matched filter) and three isolated zero
    samples."""
    iq, dib = dqpsk_iq(n, fs, seed, symbol_rate=12000.0, snr_db=20.0, freq_offset_hz=80.0)
    iq = iq.copy()
    iq[6000:6400] = 0
    iq[[10, 9000, 15001]] = 0
    return iq, dib


# --------------------------------------------------------------------------
# spectrum frames for the channel classifier (N4)
# --------------------------------------------------------------------------

def classifier_frames():
    """Synthetic spectrum frames for the classifier rows: float32 [80, 512] dB, noise at -90, a steady carrier
    (control), an intermittent one (voice), a fading one (variable) and a shoulder next to the steady one."""
    rng = np.random.default_rng(1300)
    F, N = 80, 512
    p = -90.0 + rng.standard_normal((F, N))
    p[:, 100] = -40.0 + 0.5 * rng.standard_normal(F)
    p[:, 101] = -48.0 + 0.5 * rng.standard_normal(F)
    p[:, 300] = np.where(rng.random(F) < 0.5, -45.0, -92.0) + 0.3 * rng.standard_normal(F)
    p[:, 400] = -50.0 + 6.0 * np.sin(np.arange(F) / 3.0) + 2.0 * rng.standard_normal(F)
    p[:, 200] = -82.0 + 0.5 * rng.standard_normal(F)
    freqs = ((np.arange(N) - N // 2) * (2_400_000 / N)).astype(np.float32)
    return p.astype(np.float32), freqs


# --------------------------------------------------------------------------
# dibit streams for the NID front half (N2)
# --------------------------------------------------------------------------

BCH_GENERATOR = 0o6331141367235453


def bch_encode(data16: int) -> int:
    """Systematic BCH(63,16,23) codeword (63 bits, data in bits 62..47) of a 16-bit NAC|DUID word."""
    r = data16 << 47
    for bit in range(62, 46, -1):
        if (r >> bit) & 1:
            r ^= BCH_GENERATOR << (bit - 47)
    return (data16 << 47) | r


def nid_stream(seed: int = 1400):
    """Dibit + soft-symbol stream with P25 frame heads for the framer's batch path.  In batch mode the reference
    starts the 33-dibit NID collection AT the dibit that completes the sync (p25_framer.py:503-509), so a head that
    decodes there is: 24 sync dibits whose last one doubles as NID dibit 0, NID dibits 1..10, one status dibit, NID
    dibits 11..31 ("batch" heads, NAC >= 0xC00 because the last sync dibit is 3).  "standard" heads (NID right after
    the sync, as on air) are misaligned for that path and fail BCH.  Returns (dibits uint8, soft float32, heads) with
    heads = [(kind, start index, data16, flipped bit count)]."""
    rng = np.random.default_rng(seed)
    sync = p25_sync_dibits()
    out, heads = [], []

    def gap(k):
        out.append(rng.integers(0, 4, size=k, dtype=np.uint8))

    def head(kind, data16, n_err=0, nac_err=0, second_sync_at=None):
        cw = bch_encode(data16)
        bits = [(cw >> (62 - i)) & 1 for i in range(63)] + [0]
        flips = set(int(v) for v in rng.choice(np.arange(12, 63), size=n_err, replace=False)) if n_err else set()
        flips |= set(int(v) for v in rng.choice(np.arange(2, 12), size=nac_err, replace=False)) if nac_err else set()
        for f in flips:
            bits[f] ^= 1
        nid = [(bits[2 * i] << 1) | bits[2 * i + 1] for i in range(32)]
        start = sum(len(v) for v in out)
        status = int(rng.integers(0, 4))
        if kind == "batch":
            assert nid[0] == 3
            d = list(sync) + nid[1:11] + [status] + nid[11:]
        else:
            d = list(sync) + nid[:11] + [status] + nid[11:]
        if second_sync_at is not None:
            d = d[:24 + second_sync_at] + list(sync) + nid[1:11] + [status] + nid[11:]
        out.append(np.array(d, dtype=np.uint8))
        heads.append((kind, start, data16, len(flips)))

    nac = 0xC93
    gap(40)
    for k in range(4):                                  # clean heads: the tracker locks onto the NAC
        head("batch", (nac << 4) | (k & 3) * 3)
        gap(int(rng.integers(60, 140)))
    head("standard", (0x293 << 4) | 7)
    gap(90)
    head("batch", (nac << 4) | 0x7, n_err=5)
    gap(75)
    head("batch", (nac << 4) | 0xC, n_err=11)
    gap(61)
    head("batch", (nac << 4) | 0x5, n_err=12)           # beyond t: fails
    gap(88)
    head("batch", (nac << 4) | 0x3, n_err=6, nac_err=8)  # 14 errors: only the tracked-NAC second pass recovers it
    gap(70)
    head("batch", (0xD55 << 4) | 0x0)                   # another NAC: decodes, tracker keeps the dominant one
    gap(66)
    head("batch", (nac << 4) | 0xA, second_sync_at=20)  # a second sync 20 dibits into the collection restarts it
    gap(120)
    head("batch", (nac << 4) | 0xF, n_err=2)
    gap(50)
    dib = np.concatenate(out)
    soft = (_DIBIT_TO_SYMBOL[dib] + 0.25 * rng.standard_normal(dib.size)).astype(np.float32)
    return dib, soft, heads
