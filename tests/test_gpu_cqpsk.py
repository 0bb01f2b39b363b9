"""GPU parity for row A12: Phase-2 CQPSK chain (dibits exact vs reference goldens and C oracle,
symbols within 1e-9) and the Gardner TED bank (float64, bit-exact)."""

import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu


def test_cqpsk_golden_and_oracle(golden):
    import wavehip
    from oracle.cqpsk_c import CQPSKDemodulatorRef
    from test_cqpsk_oracle import cqpsk_case

    g = golden("cqpsk")
    for ci in range(int(g["n_cases"])):
        fs, sr, iq, calls = cqpsk_case(g, ci)
        d = wavehip.CQPSKDemodulator(sample_rate=fs, symbol_rate=sr)
        dib, pos = [], 0
        for m in calls:
            dib.append(d.demodulate(iq[pos:pos + m]))
            pos += m
        assert [len(x) for x in dib] == [int(v) for v in g[f"c{ci}_counts"]], ci
        assert np.array_equal(np.concatenate(dib), g[f"c{ci}_dibits"]), ci
    # bank of 8 channels vs the C oracle, symbols too
    fs, sr, n, C = 48000, 12000, 20000, 8
    xs = np.stack([S.dqpsk_iq(n, fs, 1700 + c, symbol_rate=sr, snr_db=10.0 + 3 * c, freq_offset_hz=30.0 * c - 100)[0]
                   for c in range(C)])
    bank = wavehip.CQPSKBank(C, fs, sr, max_samples_per_call=8192)
    refs = [CQPSKDemodulatorRef(fs, sr) for _ in range(C)]
    for s in range(0, n, 8192):
        got = bank.demodulate(xs[:, s:s + 8192], want_symbols=True)
        for c in range(C):
            rd, rs = refs[c].demodulate(xs[c, s:s + 8192], want_symbols=True)
            assert np.array_equal(got[c][0], rd), (c, s)
            assert np.max(np.abs(got[c][1] - rs)) <= 1e-9 * max(1.0, np.max(np.abs(rs))), (c, s)
    bank.reset()
    again = bank.demodulate(xs[:, :4096])
    fresh = wavehip.CQPSKBank(C, fs, sr, max_samples_per_call=8192).demodulate(xs[:, :4096])
    assert all(np.array_equal(a, b) for a, b in zip(again, fresh))


def test_gardner_bit_exact(golden):
    import torch
    import wavehip
    from test_cqpsk_oracle import gardner_input

    g = golden("cqpsk")
    x = gardner_input()
    t = wavehip.GardnerTED(10.0)
    s1, e1 = t.process_block(x[:3000])
    s2, e2 = t.process_block(x[3000:])
    assert s1.dtype == np.float64 and [len(s1), len(s2)] == [int(v) for v in g["g_counts"]]
    assert np.array_equal(np.concatenate([s1, s2]), g["g_sym"]) and np.array_equal(np.concatenate([e1, e2]), g["g_err"])
    s3, e3 = wavehip.GardnerTED(10.4166666666666661).process_block(x)
    assert np.array_equal(s3, g["g2_sym"]) and np.array_equal(e3, g["g2_err"])
    # a bank: channel c gets x scaled by (c+1) -> symbols scale exactly for powers of two
    bank = wavehip.GardnerBank(4, 10.0)
    xb = torch.from_numpy(np.stack([x * s for s in (1.0, 2.0, 4.0, 0.5)])).cuda()
    sym, err, cnt = bank.process_device(xb)
    k = int(cnt[0])
    ref = wavehip.GardnerTED(10.0).process_block(x)
    assert np.array_equal(sym[0, :k].cpu().numpy(), ref[0])


def test_costas_loop_and_mueller_muller_ted_standalone(golden):
    """The two loops of the Phase-2 chain as the reference's standalone classes (CostasLoop cqpsk.py:84-196,
    MuellerMullerTED symbol_timing.py:214-380): outputs within 1e-9 of the reference golden over two calls each (carried
    state), decisions equal, `process()` one sample at a time == `process_block`, banks == single channels."""
    import torch
    import wavehip

    g = golden("cqpsk_parts")
    fs, sr, n, seed = (int(v) for v in g["args"])
    iq, _ = S.dqpsk_iq(n, fs, seed, symbol_rate=sr, snr_db=22.0, freq_offset_hz=60.0)
    x = iq.astype(np.complex128)
    cl = wavehip.CostasLoop()
    c = np.concatenate([cl.process_block(x[:3500]), cl.process_block(x[3500:])])
    assert c.dtype == np.complex128 and np.max(np.abs(c - g["costas"])) <= 1e-9
    assert abs(cl.frequency_offset - float(g["costas_freq"][0])) <= 1e-9
    one = wavehip.CostasLoop()
    assert np.max(np.abs(np.array([one.process(v) for v in x[:40]]) - g["costas"][:40])) <= 1e-9
    mm = wavehip.MuellerMullerTED(fs / sr)
    a, b = mm.process_block(g["costas"][:2500]), mm.process_block(g["costas"][2500:])
    assert [len(a[0]), len(b[0])] == [int(v) for v in g["mm_counts"]]
    assert np.max(np.abs(np.concatenate([a[0], b[0]]) - g["mm_sym"])) <= 1e-9
    assert np.max(np.abs(np.concatenate([a[1], b[1]]) - g["mm_dec"])) <= 1e-15
    assert np.max(np.abs(np.concatenate([a[2], b[2]]) - g["mm_err"])) <= 1e-9
    # banks: channel k = the stream rotated by k * 10 degrees (the loops see different phases)
    rot = np.exp(1j * np.deg2rad(10.0) * np.arange(4))[:, None]
    xb = torch.from_numpy(np.ascontiguousarray(rot * x[None, :])).cuda()
    cb = wavehip.CostasBank(4).process_device(xb)
    for k in (0, 3):
        ref = wavehip.CostasLoop().process_block((rot[k] * x))
        assert np.max(np.abs(cb[k].cpu().numpy() - ref)) <= 1e-12
    sb, db, eb, nb = wavehip.MuellerMullerBank(4, fs / sr).process_device(cb)
    s0 = wavehip.MuellerMullerTED(fs / sr).process_block(cb[0].cpu().numpy())
    assert int(nb[0]) == len(s0[0]) and np.max(np.abs(sb[0, :len(s0[0])].cpu().numpy() - s0[0])) <= 1e-12
    assert wavehip.CostasLoop().process_block(np.zeros(0)).size == 0 and len(wavehip.MuellerMullerTED(4.0).process_block([])[0]) == 0


def test_cqpsk_big_call_and_odd_samples_per_symbol(golden):
    """`cqpsk_big` (from the reference): ONE 140 000-sample demodulate() call (never split: the chain re-seeds its matched
    filter with zi * iq[0] per call, cqpsk.py:283-285) and samples_per_symbol = 5 / 5.2083, where sps / 2 is not an
    integer (the capacity bound is computed in floating point on both sides of the ABI)."""
    import wavehip
    from test_cqpsk_oracle import cqpsk_big_case

    g = golden("cqpsk_big")
    for ci in range(int(g["n_cases"])):
        fs, sr, iq, calls = cqpsk_big_case(g, ci)
        d = wavehip.CQPSKDemodulator(sample_rate=fs, symbol_rate=sr)
        dib, pos = [], 0
        for m in calls:
            dib.append(d.demodulate(iq[pos:pos + m]))
            pos += m
        assert [len(x) for x in dib] == [int(v) for v in g[f"c{ci}_counts"]], ci
        mism = np.flatnonzero(np.concatenate(dib) != g[f"c{ci}_dibits"])
        assert mism.size == 0, f"case {ci}: {mism.size} dibit mismatches, first {mism[:5]}"
    x = g["mm_in"]
    for tag in ("s5", "s52"):
        mm = wavehip.MuellerMullerTED(float(g[f"mm_{tag}_sps"][0]))
        a, b = mm.process_block(x[:2500]), mm.process_block(x[2500:])
        assert [len(a[0]), len(b[0])] == [int(v) for v in g[f"mm_{tag}_counts"]]
        assert np.max(np.abs(np.concatenate([a[0], b[0]]) - g[f"mm_{tag}_sym"])) <= 1e-9
        assert np.max(np.abs(np.concatenate([a[1], b[1]]) - g[f"mm_{tag}_dec"])) <= 1e-15
        assert np.max(np.abs(np.concatenate([a[2], b[2]]) - g[f"mm_{tag}_err"])) <= 1e-9
    with pytest.raises(Exception):
        wavehip.MuellerMullerTED(1.5)      # sps < 2: refused at create (was an integer division by zero on the host)


def test_cqpsk_muted_stretch_vs_oracle(golden):
    """Exact zeros in the input (a muted stretch longer than the matched filter, isolated zero samples): the rotated value
    of a zero sample is (+-0, +-0), its angle 0 or +-pi in the reference -- detector error 0 either way -- so the carrier
    loop coasts through it; the phase recurrence of round 3 carries that case explicitly.  Dibits equal, symbols <= 1e-9
    against the C oracle (libm atan2 with its signed-zero conventions), state carried into a second call."""
    import wavehip
    from oracle.cqpsk_c import CQPSKDemodulatorRef

    fs, sr, n = 48000, 12000, 24000
    iq, _ = S.dqpsk_muted_iq(n, fs)
    g = golden("cqpsk_big")
    assert S.sha256(iq) == str(g["muted_sha"])
    d, r = wavehip.CQPSKBank(1, fs, sr), CQPSKDemodulatorRef(fs, sr)
    got = []
    for lo, hi in ((0, 13000), (13000, n)):
        (gd, gs), = d.demodulate(iq[None, lo:hi], want_symbols=True)
        rd, rs = r.demodulate(iq[lo:hi], want_symbols=True)
        assert np.array_equal(gd, rd), (lo, np.flatnonzero(gd[:len(rd)] != rd[:len(gd)])[:5])
        assert np.max(np.abs(gs - rs)) <= 1e-9 * max(1.0, np.max(np.abs(rs)))
        got.append(gd)
    # ... and the reference itself
    assert [len(x) for x in got] == [int(v) for v in g["muted_counts"]] and np.array_equal(np.concatenate(got), g["muted_dibits"])
