"""The reference's own hot-path unit tests, re-expressed against the wavehip drop-ins (same inputs, same
assertions, our operators): backend/tests/unit/test_pack_functions.py, test_fm_demod.py, test_fft_backends.py and
backend/tests/test_p25_dsp.py / test_p25_bch.py.  The reference's tests are property tests (dtype, length, bounds,
known small answers) -- SURVEY F9 -- so they run unchanged in spirit on the device path."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wh():
    import wavehip
    return wavehip


def _wbfm(wh, iq, fs, **kw):
    return wh.process_channel_dsp_stateless(np.asarray(iq, dtype=np.complex64), fs, wh.ChannelConfig(mode="wbfm", **kw))[0]


def _nbfm(wh, iq, fs, **kw):
    kw.setdefault("enable_deemphasis", False)
    return wh.process_channel_dsp_stateless(np.asarray(iq, dtype=np.complex64), fs, wh.ChannelConfig(mode="nbfm", **kw))[0]


# ---- tests/unit/test_pack_functions.py -------------------------------------------------------------------------

def test_pack_iq16_reference_cases(wh):                                    # :15-62
    data = wh.pack_iq16(np.array([0.5 + 0.5j, -0.5 - 0.5j], dtype=np.complex64))
    assert len(data) == 8
    u = np.frombuffer(data, dtype=np.int16)
    assert abs(u[0] - 16383) < 2 and abs(u[1] - 16383) < 2 and abs(u[2] + 16383) < 2 and abs(u[3] + 16383) < 2
    assert wh.pack_iq16(np.array([], dtype=np.complex64)) == b""
    u = np.frombuffer(wh.pack_iq16(np.array([2.0 + 2.0j], dtype=np.complex64)), dtype=np.int16)
    assert abs(u[0] - 32767) < 2 and abs(u[1] - 32767) < 2
    u = np.frombuffer(wh.pack_iq16(np.array([-2.0 - 2.0j], dtype=np.complex64)), dtype=np.int16)
    assert abs(u[0] + 32767) < 2 and abs(u[1] + 32767) < 2
    u = np.frombuffer(wh.pack_iq16(np.zeros(2, dtype=np.complex64)), dtype=np.int16)
    assert np.array_equal(u, [0, 0, 0, 0])


def test_pack_pcm16_and_f32_reference_cases(wh):                           # :68-122
    u = np.frombuffer(wh.pack_pcm16(np.array([0.5, -0.5, 0.0], dtype=np.float32)), dtype=np.int16)
    assert len(u) == 3 and abs(u[0] - 16383) < 2 and abs(u[1] + 16383) < 2 and u[2] == 0
    assert wh.pack_pcm16(np.array([], dtype=np.float32)) == b""
    u = np.frombuffer(wh.pack_pcm16(np.array([2.0, -2.0], dtype=np.float32)), dtype=np.int16)
    assert abs(u[0] - 32767) < 2 and abs(u[1] + 32767) < 2
    x = np.array([0.5, -0.5, 0.0], dtype=np.float32)
    data = wh.pack_f32(x)
    assert len(data) == 12 and np.allclose(np.frombuffer(data, dtype=np.float32), x)
    assert wh.pack_f32(np.array([], dtype=np.float32)) == b""
    assert np.allclose(np.frombuffer(wh.pack_f32(np.array([2.0, -2.0], dtype=np.float32)), dtype=np.float32), [1.0, -1.0])


def test_freq_shift_properties(wh):                                        # :128-170, through the "raw" mode of the operator
    iq = np.array([1 + 0j, 0 + 1j, -1 + 0j], dtype=np.complex64)
    raw = lambda x, off, fs: wh.process_channel_dsp_stateless(x, fs, wh.ChannelConfig(mode="raw", offset_hz=off))[0]
    out = raw(iq, 0.0, 48000)
    assert np.array_equal(out.view(np.complex64), iq)                      # zero offset: unchanged
    assert wh.process_channel_dsp_stateless(np.array([], dtype=np.complex64), 48000, wh.ChannelConfig(mode="raw")) == (None, {})
    x = (0.5 * np.exp(2j * np.pi * np.arange(1000) / 37.0)).astype(np.complex64)
    y = raw(x, 1000.0, 48000).view(np.complex64)
    assert y.dtype == np.complex64 and np.allclose(np.abs(y), np.abs(x), atol=1e-5)       # magnitude preserved
    assert not np.allclose(np.angle(y[1:]), np.angle(x[1:]))                               # phase changed


# ---- tests/unit/test_fm_demod.py ---------------------------------------------------------------------------------

def test_wbfm_demod_properties(wh):                                        # :21-104
    rng = np.random.default_rng(0)
    iq = rng.standard_normal(10000).astype(np.complex64)
    audio = _wbfm(wh, iq, 200_000)
    assert audio.dtype == np.float32
    n = 20_000
    audio = _wbfm(wh, rng.standard_normal(n).astype(np.complex64) * 0.1, 200_000, audio_rate=48_000)
    expected = int(n * 48_000 / 200_000)
    assert abs(len(audio) - expected) < expected * 0.1
    audio = _wbfm(wh, rng.standard_normal(10000).astype(np.complex64) * 10, 200_000)
    assert np.max(np.abs(audio)) <= 1.0
    t = np.arange(20_000) / 200_000                                        # conftest generate_fm_signal recipe
    iq = np.exp(1j * (75_000 / 1000) * np.sin(2 * np.pi * 1000 * t)).astype(np.complex64)
    audio = _wbfm(wh, iq, 200_000, audio_rate=48_000, enable_deemphasis=False, enable_mpx_filter=False)
    assert np.std(audio) > 0.01
    assert wh.process_channel_dsp_stateless(np.array([], dtype=np.complex64), 200_000, wh.ChannelConfig(mode="wbfm")) == (None, {})
    a1 = _wbfm(wh, iq, 200_000, enable_deemphasis=True)
    a2 = _wbfm(wh, iq, 200_000, enable_mpx_filter=True, mpx_cutoff_hz=15_000)
    assert a1.dtype == np.float32 and a2.dtype == np.float32 and len(a1) == len(a2)


def test_nbfm_demod_properties(wh):                                        # :109-142
    rng = np.random.default_rng(1)
    iq = rng.standard_normal(10000).astype(np.complex64)
    assert _nbfm(wh, iq, 48_000).dtype == np.float32
    assert np.max(np.abs(_nbfm(wh, iq * 10, 48_000))) <= 1.0
    audio = _nbfm(wh, iq, 48_000, enable_fm_highpass=True, fm_highpass_hz=300, enable_fm_lowpass=True, fm_lowpass_hz=3000)
    assert audio.dtype == np.float32 and len(audio) == 10000


# ---- tests/unit/test_fft_backends.py -----------------------------------------------------------------------------

def test_fft_backend_reference_cases(wh):                                  # :35-117
    be = wh.HipFFTBackend(fft_size=2048)
    assert be.fft_size == 2048 and be.name
    w = wh.HipFFTBackend(fft_size=1024).window
    assert w.shape == (1024,) and w.dtype == np.float32 and w[0] < 0.01 and w[-1] < 0.01 and w[512] > 0.99
    fs = 48000
    iq = np.exp(2j * np.pi * 1000 * np.arange(2048) / fs).astype(np.complex64)
    r = be.execute(iq, fs)
    assert isinstance(r, wh.FFTResult) and r.power_db.shape == (2048,) and r.freqs.shape == (2048,)
    assert r.bin_hz == pytest.approx(fs / 2048, rel=0.01)
    assert abs(r.freqs[np.argmax(r.power_db)] - 1000) < 50
    r = be.execute(np.zeros(100, dtype=np.complex64), fs)                  # fewer samples than fft_size: zeros
    assert r.power_db.shape == (2048,) and np.all(r.power_db == 0)
    rng = np.random.default_rng(42)
    r = wh.HipFFTBackend(fft_size=4096).execute((rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64), fs)
    assert r.power_db.shape == (4096,) and np.std(r.power_db) < 10
    iq = (rng.standard_normal(8192) + 1j * rng.standard_normal(8192)).astype(np.complex64)
    for n in (512, 1024, 2048, 4096):
        r = wh.HipFFTBackend(fft_size=n).execute(iq, fs)
        assert r.power_db.shape == (n,) and r.freqs.shape == (n,) and r.bin_hz == pytest.approx(fs / n, rel=0.01)


# ---- tests/test_p25_dsp.py ---------------------------------------------------------------------------------------

def test_c4fm_demodulator_reference_cases(wh):                             # :148-211
    from wavehip.c4fm import design_rrc_filter
    d = wh.C4FMDemodulator(sample_rate=48000, symbol_rate=4800)
    assert d.sample_rate == 48000 and d.symbol_rate == 4800 and d.samples_per_symbol == 10.0
    rrc = design_rrc_filter(samples_per_symbol=10.0, num_taps=101, alpha=0.2)
    assert len(rrc) == 101 and abs(np.sum(rrc) - 1.0) < 1e-3 and np.sum(rrc ** 2) > 0
    dib, soft = wh.C4FMDemodulator().demodulate(np.array([], dtype=np.complex64))
    assert len(dib) == 0 and len(soft) == 0
    rng = np.random.default_rng(2)
    tx = rng.integers(0, 4, 100)
    freq = np.repeat(np.array([600, 1800, -600, -1800])[tx], 10)
    iq = np.exp(1j * np.cumsum(2 * np.pi * freq / 48000)).astype(np.complex64)
    dib, soft = d.demodulate(iq)
    assert len(dib) > 0 and len(soft) == len(dib)
    d.demodulate((rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64))
    d.reset()
    assert d._ted_phase == 0.0


def test_cqpsk_and_ted_reference_cases(wh):                                # :217-312
    d = wh.CQPSKDemodulator(sample_rate=48000, symbol_rate=12000)
    assert d.sample_rate == 48000 and d.symbol_rate == 12000 and d.samples_per_symbol == 4.0
    assert len(wh.CQPSKDemodulator().demodulate(np.array([], dtype=np.complex64))) == 0
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64)
    a = d.demodulate(x)
    d.reset()
    assert np.array_equal(d.demodulate(x), a)                              # reset restores the initial state
    ted = wh.GardnerTED(samples_per_symbol=10.0)
    assert ted.samples_per_symbol == 10.0
    s1, e1 = ted.process_block(rng.standard_normal(100))
    ted.reset()
    s2, e2 = ted.process_block(np.zeros(100))
    assert len(s2) == len(e2) and not np.any(s2)


def test_lsm_demodulator_like_reference_cqpsk_sample(wh):                  # tests/test_cqpsk_sample.py shape: stream in chunks
    d = wh.LSMDemodulator(sample_rate=48000, symbol_rate=4800)
    assert d.samples_per_symbol == 10.0
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(20000) + 1j * rng.standard_normal(20000)).astype(np.complex64) * 0.1
    total = sum(len(d.demodulate(x[i:i + 2000])) for i in range(0, 20000, 2000))
    assert 0.9 * 2000 <= total <= 1.2 * 2000


def test_bch_reference_cases(wh):                                          # tests/test_p25_bch.py:38-56
    assert wh.bch_decode(np.zeros(63, dtype=np.uint8)) == (0, 0)
    data, errors = wh.bch_decode(np.zeros(63, dtype=np.uint8), 0x123)
    assert errors >= 0 or errors == -1
    tr = wh.NACTracker()                                                   # :62-80
    assert tr.get_tracked_nac() == 0
    tr.track(0x123); assert tr.get_tracked_nac() == 0
    tr.track(0x123); assert tr.get_tracked_nac() == 0
    tr.track(0x123); assert tr.get_tracked_nac() == 0x123
