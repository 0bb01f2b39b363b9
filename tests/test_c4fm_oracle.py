"""Pin the C oracle of the C4FM demodulator (oracle/c4fm_ref.c) to the goldens captured from the
reference (tests/golden/c4fm.npz): dibits bit-exact, soft symbols <= 1e-5 peak-relative, for
both atan2 flavours (libm and the portable polynomial shared with the HIP kernel)."""

import numpy as np
import pytest

import signals as S
from conftest import peak_rel_err
from oracle.c4fm_c import C4FMDemodulatorRef, design_baseband_lpf, design_rrc_filter


def _case(g, ci):
    fs, n, call, seed, snr10, foff10, s0, s1 = (int(v) for v in g[f"c{ci}_args"])
    sil = None if s0 < 0 else (s0, s1)
    iq, _ = S.c4fm_iq(n, fs, seed, snr_db=snr10 / 10.0, freq_offset_hz=foff10 / 10.0, silence=sil)
    assert S.sha256(iq) == str(g[f"c{ci}_sha"])
    return fs, n, call, iq


def test_filter_designs_match_reference(golden):
    g = golden("c4fm")
    for fs in (48000, 50000, 19200):
        sps = fs / 4800
        assert np.array_equal(design_baseband_lpf(fs), g[f"lpf_{fs}"])
        assert np.array_equal(design_rrc_filter(sps, int(16 * sps) + 1), g[f"rrc_{fs}"])


@pytest.mark.parametrize("atan_mode", [0, 1])
def test_c4fm_oracle_matches_reference(golden, atan_mode):
    g = golden("c4fm")
    for ci in range(int(g["n_cases"])):
        fs, n, call, iq = _case(g, ci)
        d = C4FMDemodulatorRef(sample_rate=fs, atan_mode=atan_mode)
        dib, soft, counts = [], [], []
        for s in range(0, n, call):
            a, b = d.demodulate(iq[s:s + call])
            dib.append(a); soft.append(b); counts.append(len(a))
        dib, soft = np.concatenate(dib), np.concatenate(soft)
        assert np.array_equal(np.array(counts, dtype=np.int32), g[f"c{ci}_counts"]), ci
        ref_d, ref_s = g[f"c{ci}_dibits"], g[f"c{ci}_soft"]
        mism = np.flatnonzero(dib != ref_d)
        assert mism.size == 0, f"case {ci}: {mism.size} dibit mismatches, first at {mism[:5]}"
        assert peak_rel_err(soft, ref_s) <= 1e-5, ci
        st = d.state()
        ref_state = g[f"c{ci}_state"]
        assert st["sync_count"] == int(ref_state[0]) and st["fine_sync"] == bool(ref_state[1])
        assert abs(st["pll"] - ref_state[2]) <= 1e-5 and abs(st["gain"] - ref_state[3]) <= 1e-6
        assert abs(st["sample_point"] - ref_state[4]) <= 1e-6


def test_c4fm_oracle_reset_and_empty(golden):
    g = golden("c4fm")
    fs, n, call, iq = _case(g, 3)
    d = C4FMDemodulatorRef(sample_rate=fs)
    a1, s1 = d.demodulate(iq[:call * 3])
    d.reset()
    a2, s2 = d.demodulate(iq[:call * 3])
    assert np.array_equal(a1, a2) and np.array_equal(s1, s2)
    e = d.demodulate(np.empty(0, np.complex64))
    assert e[0].size == 0 and e[1].size == 0


@pytest.mark.parametrize("atan_mode", [0, 1])
def test_c4fm_oracle_matches_reference_on_the_frame_head_carrier(golden, atan_mode):
    """The config-4 chain golden (IQ of a carrier transmitting frame heads -> reference demodulator -> reference framer):
    the C oracle reproduces the reference demodulator's dibits and soft symbols on both cases, 100 ms calls."""
    g = golden("chain4")
    for ci in range(int(g["n_cases"])):
        fs, call, seed, snr10, foff10, reps = (int(v) for v in g[f"c{ci}_args"])
        iq, _ = S.p25_head_stream_iq(fs, seed, reps, snr10 / 10.0, foff10 / 10.0)
        assert S.sha256(iq) == str(g[f"c{ci}_sha"])
        d = C4FMDemodulatorRef(sample_rate=fs, atan_mode=atan_mode)
        dib, soft, counts = [], [], []
        for s in range(0, len(iq), call):
            a, b = d.demodulate(iq[s:s + call])
            dib.append(a); soft.append(b); counts.append(len(a))
        assert np.array_equal(np.array(counts, dtype=np.int32), g[f"c{ci}_counts"]), ci
        assert np.array_equal(np.concatenate(dib), g[f"c{ci}_dibits"]), ci
        assert peak_rel_err(np.concatenate(soft), g[f"c{ci}_soft"]) <= 1e-5, ci


def big_case(g, ci):
    fs, seed, snr10, foff10 = (int(v) for v in g[f"c{ci}_args"])
    calls = [int(v) for v in g[f"c{ci}_calls"]]
    iq, _ = S.c4fm_iq(sum(calls), fs, seed, snr_db=snr10 / 10.0, freq_offset_hz=foff10 / 10.0)
    assert S.sha256(iq) == str(g[f"c{ci}_sha"])
    return fs, calls, iq


@pytest.mark.parametrize("atan_mode", [0, 1])
def test_c4fm_oracle_matches_reference_at_production_call_sizes(golden, atan_mode):
    """`c4fm_big`: demodulate() at the callers' sizes -- single 72 000 / 75 000-sample calls (trunking/system.py:1548-1549),
    three consecutive ones, one 200 000-sample call (two phase-buffer shifts, early symbols fall out of the buffer:
    index -1, c4fm.py:716-728), 150 000 samples at sps = 4, mixed small / large calls.  The reference is not
    cut-invariant, so every call is made exactly as the golden's was."""
    g = golden("c4fm_big")
    for ci in range(int(g["n_cases"])):
        fs, calls, iq = big_case(g, ci)
        d = C4FMDemodulatorRef(sample_rate=fs, atan_mode=atan_mode)
        dib, soft, counts, pos = [], [], [], 0
        for m in calls:
            a, b = d.demodulate(iq[pos:pos + m])
            pos += m
            dib.append(a); soft.append(b); counts.append(len(a))
        assert np.array_equal(np.array(counts, dtype=np.int32), g[f"c{ci}_counts"]), ci
        mism = np.flatnonzero(np.concatenate(dib) != g[f"c{ci}_dibits"])
        assert mism.size == 0, f"case {ci}: {mism.size} dibit mismatches, first at {mism[:5]}"
        assert peak_rel_err(np.concatenate(soft), g[f"c{ci}_soft"]) <= 1e-5, ci
        st, ref = d.state(), g[f"c{ci}_state"]
        assert st["sync_count"] == int(ref[0]) and st["fine_sync"] == bool(ref[1]) and st["buffer_pointer"] == int(ref[5])
        assert abs(st["pll"] - ref[2]) <= 1e-5 and abs(st["gain"] - ref[3]) <= 1e-6 and abs(st["sample_point"] - ref[4]) <= 1e-6


@pytest.mark.parametrize("atan_mode", [0, 1])
def test_c4fm_oracle_muted_input(golden, atan_mode):
    """Exact zeros in the input (`c4fm_big` muted cases, from the reference): zeros pass both FIRs exactly, the
    discriminator's conjugate product is made of signed zeros and np.arctan2 returns 0 or +-pi by their signs -- both atan2
    flavours of the oracle follow IEEE there (the portable one took the quadrant from `x < 0` until round 3 and returned 0
    for (+0, -0): one dibit of 9 600 off on this stream)."""
    g = golden("c4fm_big")
    iq, _ = S.c4fm_muted_iq()
    assert S.sha256(iq) == str(g["muted_sha"])
    for tag in ("muted_a", "muted_b"):
        d = C4FMDemodulatorRef(sample_rate=48000, atan_mode=atan_mode)
        dib, soft, counts, pos = [], [], [], 0
        for m in (int(v) for v in g[f"{tag}_calls"]):
            a, b = d.demodulate(iq[pos:pos + m])
            pos += m
            dib.append(a); soft.append(b); counts.append(len(a))
        assert np.array_equal(np.array(counts, dtype=np.int32), g[f"{tag}_counts"]), tag
        assert np.array_equal(np.concatenate(dib), g[f"{tag}_dibits"]), tag
        assert peak_rel_err(np.concatenate(soft), g[f"{tag}_soft"]) <= 1e-5, tag
