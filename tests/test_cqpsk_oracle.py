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


def test_costas_and_mueller_muller_oracle_match_reference(golden):
    """The numpy restatements of CostasLoop (cqpsk.py:84-196) and MuellerMullerTED (symbol_timing.py:214-380) reproduce
    the reference's outputs on the `cqpsk_parts` golden (two calls each, carried state)."""
    from oracle import ref_np as O

    g = golden("cqpsk_parts")
    fs, sr, n, seed = (int(v) for v in g["args"])
    iq, _ = S.dqpsk_iq(n, fs, seed, symbol_rate=sr, snr_db=22.0, freq_offset_hz=60.0)
    assert S.sha256(iq) == str(g["sha"])
    x = iq.astype(np.complex128)
    cl = O.CostasLoop()
    c = np.concatenate([cl.process_block(x[:3500]), cl.process_block(x[3500:])])
    assert np.max(np.abs(c - g["costas"])) <= 1e-12 and abs(cl._freq - float(g["costas_freq"][0])) <= 1e-14
    mm = O.MuellerMullerTED(fs / sr)
    a, b = mm.process_block(g["costas"][:2500]), mm.process_block(g["costas"][2500:])
    assert [len(a[0]), len(b[0])] == [int(v) for v in g["mm_counts"]]
    assert np.max(np.abs(np.concatenate([a[0], b[0]]) - g["mm_sym"])) <= 1e-12
    assert np.array_equal(np.concatenate([a[1], b[1]]), g["mm_dec"])
    assert np.max(np.abs(np.concatenate([a[2], b[2]]) - g["mm_err"])) <= 1e-12


def cqpsk_big_case(g, ci):
    fs, sr, seed, snr10, foff10 = (int(v) for v in g[f"c{ci}_args"])
    calls = [int(v) for v in g[f"c{ci}_calls"]]
    iq, _ = S.dqpsk_iq(sum(calls), fs, seed, symbol_rate=sr, snr_db=snr10 / 10.0, freq_offset_hz=foff10 / 10.0)
    assert S.sha256(iq) == str(g[f"c{ci}_sha"])
    return fs, sr, iq, calls


def test_cqpsk_oracle_matches_reference_big_call_and_odd_sps(golden):
    """`cqpsk_big`: one 140 000-sample call (the chain re-seeds its matched filter with zi * iq[0] per call,
    cqpsk.py:283-285: results depend on the cut), samples_per_symbol 5 and 5.2083 (sps / 2 not an integer)."""
    g = golden("cqpsk_big")
    for ci in range(int(g["n_cases"])):
        fs, sr, iq, calls = cqpsk_big_case(g, ci)
        d = CQPSKDemodulatorRef(sample_rate=fs, symbol_rate=sr)
        dib, pos = [], 0
        for m in calls:
            dib.append(d.demodulate(iq[pos:pos + m]))
            pos += m
        assert [len(x) for x in dib] == [int(v) for v in g[f"c{ci}_counts"]], ci
        mism = np.flatnonzero(np.concatenate(dib) != g[f"c{ci}_dibits"])
        assert mism.size == 0, f"case {ci}: {mism.size} dibit mismatches, first {mism[:5]}"


def test_mueller_muller_oracle_odd_sps(golden):
    from oracle import ref_np as O

    g = golden("cqpsk_big")
    x = g["mm_in"]
    for tag in ("s5", "s52"):
        mm = O.MuellerMullerTED(float(g[f"mm_{tag}_sps"][0]))
        a, b = mm.process_block(x[:2500]), mm.process_block(x[2500:])
        assert [len(a[0]), len(b[0])] == [int(v) for v in g[f"mm_{tag}_counts"]]
        assert np.max(np.abs(np.concatenate([a[0], b[0]]) - g[f"mm_{tag}_sym"])) <= 1e-12
        assert np.array_equal(np.concatenate([a[1], b[1]]), g[f"mm_{tag}_dec"])
        assert np.max(np.abs(np.concatenate([a[2], b[2]]) - g[f"mm_{tag}_err"])) <= 1e-12


def test_cqpsk_oracle_muted_stretch(golden):
    """Exact zeros in the input (`cqpsk_big` muted case, from the reference): the C oracle's libm atan2 follows the same
    signed-zero conventions as np.angle, dibits equal over two calls."""
    g = golden("cqpsk_big")
    iq, _ = S.dqpsk_muted_iq()
    assert S.sha256(iq) == str(g["muted_sha"])
    d = CQPSKDemodulatorRef(sample_rate=48000, symbol_rate=12000)
    parts = [d.demodulate(iq[:13000]), d.demodulate(iq[13000:])]
    assert [len(x) for x in parts] == [int(v) for v in g["muted_counts"]]
    assert np.array_equal(np.concatenate(parts), g["muted_dibits"])
