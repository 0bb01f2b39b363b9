"""GPU parity for row A12 (LSM): the P25 Phase-1 CQPSK/LSM demodulator of decoders/p25.py:190-669.
Dibits bit-exact vs the reference goldens; dibits, per-symbol phases and carried state BIT-IDENTICAL vs the
C oracle's portable flavour (same single-rounding operations in the same order on both sides), on single
channels and on a 64-channel bank over a long ragged stream."""

import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu

STATE_KEYS = ("agc_gain", "freq_offset", "phase_acc", "symbol_clock", "prev_symbol")


def test_lsm_golden_and_oracle(golden):
    import wavehip
    from wavehip import lsm
    from oracle.lsm_c import LSMDemodulatorRef
    from test_lsm_oracle import lsm_case

    g = golden("lsm")
    assert np.array_equal(lsm.generate_mmse_taps(), g["mmse"])
    for ci in range(int(g["n_cases"])):
        fs, sr, x, lens = lsm_case(g, ci)
        assert np.array_equal(lsm.design_baseband_filter(fs), g[f"c{ci}_lpf"])
        d = wavehip.LSMDemodulator(fs, sr)
        r = LSMDemodulatorRef(fs, sr, 1)
        dib, ph, pos = [], [], 0
        for ln in lens:
            o = d.demodulate(x[pos:pos + ln])
            ro = r.demodulate(x[pos:pos + ln])
            pos += ln
            assert o.dtype == np.uint8 and np.array_equal(o, ro), (ci, pos)
            assert d.last_phases.tobytes() == r.last_phases.tobytes(), (ci, pos)
            sd, sr_ = d.state(), r.state()
            for k in STATE_KEYS:
                assert sd[k] == sr_[k], (ci, pos, k, sd[k], sr_[k])
            assert sd["clock_is_f32"] == sr_["f32mode"]
            dib.append(o)
            ph.append(d.last_phases)
        assert [len(v) for v in dib] == [int(v) for v in g[f"c{ci}_counts"]], ci
        assert np.array_equal(np.concatenate(dib), g[f"c{ci}_dibits"]), ci
        dphi = np.abs(np.concatenate(ph).astype(np.float64) - g[f"c{ci}_phases"])
        dphi = np.minimum(dphi, 2 * np.pi - dphi)
        assert dphi[:500].max() <= 2e-6 and dphi.max() <= 1e-2, (ci, dphi.max())
        assert d.state()["clock_is_f32"] == bool(g[f"c{ci}_clock_is_f32"])


def test_lsm_bank_64_channels_bit_identical_to_oracle():
    """64 channels (different payloads, SNRs 8..30 dB, offsets -120..+120 Hz, one silent, one all-zero),
    60 000 samples each in ragged calls (incl. < 63-sample calls that skip the low-pass): every channel's
    dibits, phases and state equal the C oracle's portable flavour bit for bit -- far beyond the ~5 000-symbol
    horizon over which the reference itself is reproducible across math libraries."""
    import wavehip
    from oracle.lsm_c import LSMDemodulatorRef

    fs, sr, n, C = 48000, 4800, 60000, 64
    xs = np.stack([S.dqpsk_iq(n, fs, 1900 + c, symbol_rate=sr, snr_db=8.0 + (c % 12) * 2,
                              freq_offset_hz=(c - 32) * 3.75)[0] for c in range(C)])
    xs[5, 20000:26000] = 0
    xs[6] = 0
    rng = np.random.default_rng(1999)
    lens = []
    while sum(lens) < n:
        lens.append(int(rng.integers(700, 2400)))
        if len(lens) % 5 == 3:
            lens.append(int(rng.integers(1, 62)))
    lens[-1] -= sum(lens) - n
    lens = [v for v in lens if v > 0]
    assert sum(lens) == n
    bank = wavehip.LSMBank(C, fs, sr, max_samples_per_call=4096)
    refs = [LSMDemodulatorRef(fs, sr, 1) for _ in range(C)]
    pos, total = 0, 0
    for ln in lens:
        dib, ph = bank.demodulate(xs[:, pos:pos + ln], want_phases=True)
        for c in range(C):
            ro = refs[c].demodulate(xs[c, pos:pos + ln])
            assert np.array_equal(dib[c], ro), (c, pos)
            assert ph[c].tobytes() == refs[c].last_phases.tobytes(), (c, pos)
            total += len(ro)
        pos += ln
    for c in range(C):
        sd, sr_ = bank.state(c), refs[c].state()
        for k in STATE_KEYS:
            assert sd[k] == sr_[k], (c, k, sd[k], sr_[k])
    assert total > 0.9 * C * n / (fs / sr)
    bank.reset()
    again = bank.demodulate(xs[:, :lens[0]])
    fresh = LSMDemodulatorRef(fs, sr, 1).demodulate(xs[3, :lens[0]])
    assert np.array_equal(again[3], fresh)


def test_lsm_bank_lane_form_equals_wave_form():
    """Beyond 2048 channels the feedback kernel runs one LANE per channel instead of one wave (lsm.hip): same arithmetic, so
    a 2112-channel bank (33 copies of 64 streams) returns, channel by channel, exactly what the 64-channel bank -- pinned to
    the oracle above -- returns, over three calls with carried state."""
    import wavehip

    fs, sr, n, C, reps = 48000, 4800, 9000, 64, 33
    xs = np.stack([S.dqpsk_iq(n, fs, 2100 + c, symbol_rate=sr, snr_db=10.0 + (c % 10) * 2,
                              freq_offset_hz=(c - 32) * 3.0)[0] for c in range(C)])
    small = wavehip.LSMBank(C, fs, sr, max_samples_per_call=4096)
    big = wavehip.LSMBank(C * reps, fs, sr, max_samples_per_call=4096)
    xb = np.tile(xs, (reps, 1))
    for lo, hi in ((0, 3000), (3000, 3050), (3050, 7000)):
        d0, p0 = small.demodulate(xs[:, lo:hi], want_phases=True)
        d1, p1 = big.demodulate(xb[:, lo:hi], want_phases=True)
        for c in range(C * reps):
            assert np.array_equal(d1[c], d0[c % C]), (c, lo)
            assert p1[c].tobytes() == p0[c % C].tobytes(), (c, lo)
    assert big.state(C * reps - 1) == small.state(C - 1)


def test_lsm_edges():
    import wavehip

    d = wavehip.LSMDemodulator(48000, 4800)
    assert d.sample_rate == 48000 and d.symbol_rate == 4800 and d.samples_per_symbol == 10.0
    e = d.demodulate(np.array([], dtype=np.complex64))
    assert e.dtype == np.uint8 and e.size == 0
    assert d.demodulate(np.zeros(7, dtype=np.float32)).size == 0            # odd-length real input (p25.py:427-432)
    x = S.dqpsk_iq(4000, 48000, 1950, symbol_rate=4800)[0]
    inter = np.ascontiguousarray(x).view(np.float32).astype(np.float64)     # even-length real input = interleaved IQ
    a = d.demodulate(inter)
    d2 = wavehip.LSMDemodulator(48000, 4800)
    assert np.array_equal(a, d2.demodulate(x))
    # a call longer than the bank was created for grows the work buffers (the reference takes any length and is not
    # cut-invariant, so the call is never split) and keeps the carried state: same dibits, phases and state as a bank
    # created large enough, on a stream of three calls of which the second outgrows the small bank twice over
    xs = np.stack([S.dqpsk_iq(9000, 48000, 1960 + c, symbol_rate=4800, snr_db=14.0)[0] for c in range(2)])
    small = wavehip.LSMBank(2, 48000, 4800, max_samples_per_call=128)
    big = wavehip.LSMBank(2, 48000, 4800, max_samples_per_call=16384)
    for lo, hi in ((0, 100), (100, 5000), (5000, 9000)):
        (da, pa), (db, pb) = small.demodulate(xs[:, lo:hi], want_phases=True), big.demodulate(xs[:, lo:hi], want_phases=True)
        for c in range(2):
            assert np.array_equal(da[c], db[c]) and pa[c].tobytes() == pb[c].tobytes(), (lo, c)
    assert small.state(1) == big.state(1) and small.max_samples_per_call >= 4900
