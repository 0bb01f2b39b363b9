"""GPU parity of the P25 C4FM bank (HIP, through the C ABI): dibits BIT-EXACT against the
reference goldens and against the C oracle (portable atan2 flavour == the kernel's), soft
symbols within 1e-5 peak-relative."""

import numpy as np
import pytest

import signals as S
from conftest import peak_rel_err

pytestmark = pytest.mark.gpu


def _case(g, ci):
    fs, n, call, seed, snr10, foff10, s0, s1 = (int(v) for v in g[f"c{ci}_args"])
    sil = None if s0 < 0 else (s0, s1)
    iq, _ = S.c4fm_iq(n, fs, seed, snr_db=snr10 / 10.0, freq_offset_hz=foff10 / 10.0, silence=sil)
    return fs, n, call, iq


def test_c4fm_golden_streamed_calls(golden):
    import wavehip

    g = golden("c4fm")
    for ci in range(int(g["n_cases"])):
        fs, n, call, iq = _case(g, ci)
        d = wavehip.C4FMDemodulator(sample_rate=fs)
        dib, soft, counts = [], [], []
        for s in range(0, n, call):
            a, b = d.demodulate(iq[s:s + call])
            assert a.dtype == np.uint8 and b.dtype == np.float32
            dib.append(a); soft.append(b); counts.append(len(a))
        dib, soft = np.concatenate(dib), np.concatenate(soft)
        assert np.array_equal(np.array(counts, dtype=np.int32), g[f"c{ci}_counts"]), ci
        mism = np.flatnonzero(dib != g[f"c{ci}_dibits"])
        assert mism.size == 0, f"case {ci}: {mism.size} dibit mismatches vs reference, first {mism[:5]}"
        assert peak_rel_err(soft, g[f"c{ci}_soft"]) <= 1e-5, ci


def test_c4fm_bank_bit_exact_vs_c_oracle():
    """16 channels (different seeds / SNR / offsets, one of them noise only), 100 ms calls, 3 s:
    dibits AND soft symbols bit-identical to the C oracle run with the same portable atan2."""
    import wavehip
    from oracle.c4fm_c import C4FMDemodulatorRef

    fs, n, call, C = 48000, 48000 * 3, 4800, 16
    rng = np.random.default_rng(5)
    iqs = []
    for c in range(C):
        snr = -30.0 if c == 7 else float(rng.uniform(6, 30))
        iq, _ = S.c4fm_iq(n, fs, 2000 + c, snr_db=snr, freq_offset_hz=float(rng.uniform(-400, 400)))
        iqs.append(iq)
    x = np.stack(iqs)
    bank = wavehip.C4FMBank(C, fs, max_samples_per_call=call)
    refs = [C4FMDemodulatorRef(sample_rate=fs, atan_mode=1) for _ in range(C)]
    tot = 0
    for s in range(0, n, call):
        got = bank.demodulate(x[:, s:s + call])
        for c in range(C):
            rd, rs = refs[c].demodulate(x[c, s:s + call])
            assert np.array_equal(got[c][0], rd), (c, s)
            assert np.array_equal(got[c][1], rs), (c, s)
            tot += len(rd)
    assert tot > C * 14000
    assert max(r.state()["sync_count"] for r in refs) > 20
    # reset() == fresh demodulator
    bank.reset()
    again = bank.demodulate(x[:, :call])
    fresh = wavehip.C4FMBank(C, fs, max_samples_per_call=call).demodulate(x[:, :call])
    for c in range(C):
        assert np.array_equal(again[c][0], fresh[c][0]) and np.array_equal(again[c][1], fresh[c][1])


def test_c4fm_rates_and_ragged_calls():
    """50 kHz (fractional sps) and 19.2 kHz (sps = 4), ragged call lengths incl. tiny ones."""
    import wavehip
    from oracle.c4fm_c import C4FMDemodulatorRef

    for fs, seed in ((50000, 31), (19200, 32)):
        n = fs * 2
        iq, _ = S.c4fm_iq(n, fs, seed, snr_db=18.0, freq_offset_hz=150.0)
        d = wavehip.C4FMDemodulator(sample_rate=fs)
        r = C4FMDemodulatorRef(sample_rate=fs, atan_mode=1)
        pos, lens = 0, [1, 7, 63, 500, 4097, 9999, 3, 12000]
        i = 0
        while pos < n:
            m = min(lens[i % len(lens)], n - pos)
            a, b = d.demodulate(iq[pos:pos + m])
            ra, rb = r.demodulate(iq[pos:pos + m])
            assert np.array_equal(a, ra) and np.array_equal(b, rb), (fs, pos, m)
            pos += m
            i += 1
    e = wavehip.C4FMDemodulator(sample_rate=48000).demodulate(np.empty(0, np.complex64))
    assert e[0].size == 0 and e[1].size == 0
