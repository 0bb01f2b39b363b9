"""Pin the C oracle of row A12 (Phase-2 CQPSK chain + GardnerTED) to goldens from the reference."""

import numpy as np

import signals as S
from oracle.cqpsk_c import CQPSKDemodulatorRef, GardnerTEDRef


def gardner_input():
    rng = np.random.default_rng(1600)
    sym = rng.choice([-3.0, -1.0, 1.0, 3.0], size=700)
    x = np.repeat(sym, 10).astype(np.float64)
    x = np.convolve(x, np.hanning(15) / np.sum(np.hanning(15)), mode="same") + 0.05 * rng.standard_normal(x.size)
    return x.astype(np.float32)


def cqpsk_case(g, ci):
    fs, sr, n, seed, snr10, foff10 = (int(v) for v in g[f"c{ci}_args"])
    iq, _ = S.dqpsk_iq(n, fs, seed, symbol_rate=sr, snr_db=snr10 / 10.0, freq_offset_hz=foff10 / 10.0)
    assert S.sha256(iq) == str(g[f"c{ci}_sha"])
    return fs, sr, iq, [int(v) for v in g[f"c{ci}_calls"]]


def test_cqpsk_oracle_matches_reference(golden):
    g = golden("cqpsk")
    for ci in range(int(g["n_cases"])):
        fs, sr, iq, calls = cqpsk_case(g, ci)
        d = CQPSKDemodulatorRef(sample_rate=fs, symbol_rate=sr)
        dib, pos = [], 0
        for m in calls:
            dib.append(d.demodulate(iq[pos:pos + m]))
            pos += m
        assert [len(x) for x in dib] == [int(v) for v in g[f"c{ci}_counts"]], ci
        dib = np.concatenate(dib)
        mism = np.flatnonzero(dib != g[f"c{ci}_dibits"])
        assert mism.size == 0, f"case {ci}: {mism.size} dibit mismatches, first {mism[:5]}"


def test_gardner_oracle_matches_reference(golden):
    g = golden("cqpsk")
    x = gardner_input()
    assert S.sha256(x) == str(g["g_sha"])
    t = GardnerTEDRef(10.0)
    s1, e1 = t.process_block(x[:3000])
    s2, e2 = t.process_block(x[3000:])
    assert [len(s1), len(s2)] == [int(v) for v in g["g_counts"]]
    assert np.array_equal(np.concatenate([s1, s2]), g["g_sym"])          # float64, bit-exact
    assert np.array_equal(np.concatenate([e1, e2]), g["g_err"])
    s3, e3 = GardnerTEDRef(10.4166666666666661).process_block(x)
    assert np.array_equal(s3, g["g2_sym"]) and np.array_equal(e3, g["g2_err"])
    t.reset()
    s4, _ = t.process_block(x[:3000])
    assert np.array_equal(s4, s1)
