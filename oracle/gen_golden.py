#!/usr/bin/env python3
"""Generate golden vectors by importing the real WaveCap-SDR reference.

Runs ONLY in the build container (needs /root/reference); the GPU box never sees
the reference.  Output: small ``tests/golden/*.npz`` files holding recipe
arguments (see ``tests/signals.py``), a sha256 of each generated input and the
reference's outputs.  Usage:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/oracle/gen_golden.py [names...]

Oracle semantics recorded in every fixture: numpy / scipy versions, numba absent
(SURVEY.md F10).
"""

from __future__ import annotations

import os
import sys

REF = "/root/reference/backend"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402
import scipy  # noqa: E402

import signals as S  # noqa: E402

import wavecapsdr.trunking  # noqa: E402,F401  (must precede capture: circular import)
from wavecapsdr import capture as rc  # noqa: E402
from wavecapsdr.dsp import fm as rfm  # noqa: E402
from wavecapsdr.dsp.channelizer import PolyphaseChannelizer  # noqa: E402
from wavecapsdr.dsp.fft.scipy_backend import ScipyFFTBackend  # noqa: E402
from wavecapsdr.dsp.p25 import c4fm as rc4  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
META = dict(numpy=np.__version__, scipy=scipy.__version__, numba=bool(rc4.NUMBA_AVAILABLE))


def save(name, **kw):
    kw["meta"] = np.array(repr(META))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def gen_a1():
    i16 = np.arange(-32768, 32768, dtype=np.int16)
    # reference unpack rule cli.py:447-452
    unpack = i16.astype(np.float32) / 32768.0
    rng = np.random.default_rng(11)
    edge = np.array([1.0, -1.0, 1.0 + 1e-7, -1.0 - 1e-7, 0.5, -0.5, 0.99999, -0.99999, 0.0, -0.0,
                     1.5, -1.5, 3.0517578e-05, -3.0517578e-05, 1e-9, 0.999984], dtype=np.float32)
    vals = np.concatenate([edge, rng.uniform(-1.2, 1.2, 8192 - edge.size).astype(np.float32)])
    cin = (vals[0::2] + 1j * vals[1::2]).astype(np.complex64)
    packed = np.frombuffer(rc.pack_iq16(cin.copy()), dtype=np.int16)
    pcm = np.frombuffer(rc.pack_pcm16(vals.copy()), dtype=np.int16)
    save("a1_int16", i16=i16, unpack_f32=unpack, pack_in=cin, pack_out=packed, pcm_in=vals, pcm_out=pcm)


def gen_a2():
    cases = [(120000, 25000, 2400000), (120000, -25000, 2400000), (120000, 1, 2400000),
             (8192, 1199999, 2400000), (500000, 4900000, 10000000), (120000, -775000, 2400000)]
    out = {}
    for ci, (n, off, fs) in enumerate(cases):
        iq = S.noise_c64(n, 200 + ci)
        rc._get_freq_shift_exp.cache_clear()
        y = rc.freq_shift(iq, float(off), fs)
        rng = np.random.default_rng(300 + ci)
        idx = np.unique(np.concatenate([np.arange(256), np.arange(n - 256, n),
                                        rng.integers(0, n, 1024)])).astype(np.int64)
        out[f"c{ci}_args"] = np.array([n, off, fs, 200 + ci], dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(iq))
        out[f"c{ci}_idx"] = idx
        out[f"c{ci}_y"] = y[idx]
        out[f"c{ci}_tab"] = rc._get_freq_shift_exp(n, off, fs)[idx]
    out["n_cases"] = np.array(len(cases))
    save("a2_nco", **out)


def gen_a3():
    fs = 2400000
    iq = S.fm_tone_c64(8192, fs, seed=31, audio_hz=1000.0, deviation=75000.0)
    y = rfm.quadrature_demod(iq, fs)
    iq2 = S.noise_c64(4096, 32)
    y2 = rfm.quadrature_demod(iq2, 250000)
    x = (np.random.default_rng(33).standard_normal(4096) * 0.7).astype(np.float32)
    save("a3_quad", iq=iq, fs=np.array(fs), y=y, iq2=iq2, fs2=np.array(250000), y2=y2,
         x=x, soft_clip=rfm.soft_clip(x), rms_norm=rfm.rms_normalize(x, target_rms=0.18))


def gen_a6():
    out = {}
    cases = [(2400000, 48000, 120000, 61), (1000000, 48000, 20000, 62), (10000000, 48000, 50000, 63),
             (2400000, 48000, 12345, 64), (250000, 48000, 8000, 65)]
    for ci, (fi, fo, n, seed) in enumerate(cases):
        x = (np.random.default_rng(seed).standard_normal(n) * 0.3).astype(np.float32)
        y = rfm.resample_poly(x, fi, fo)
        out[f"c{ci}_args"] = np.array([fi, fo, n, seed], dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(x))
        out[f"c{ci}_y"] = y
    out["n_cases"] = np.array(len(cases))
    save("a6_resample", **out)


def _cfg(mode, off):
    cfg = rc.ChannelConfig(id="g", capture_id="c", mode=mode, offset_hz=float(off))
    if mode == "nbfm":  # mode defaults, capture.py:3444-3452
        cfg.enable_deemphasis = False
        cfg.enable_mpx_filter = False
        cfg.enable_fm_highpass = False
        cfg.enable_fm_lowpass = False
    return cfg


def gen_chain():
    out = {}
    fs = 2400000
    n = 120000
    # NBFM: config-2 content, two consecutive chunks, three channel offsets
    offs = S.nbfm_bank_offsets()
    k_list = [0, 13, 31]
    for chunk in range(2):
        iq = S.nbfm_bank_c64(n, fs, seed=500 + chunk, start=chunk * n)
        i16 = S.pack_iq16_np(iq)
        iq_q = i16.astype(np.float32) / 32768.0
        iq_q = (iq_q[0::2] + 1j * iq_q[1::2]).astype(np.complex64)
        out[f"nbfm{chunk}_sha_i16"] = np.array(S.sha256(i16))
        for k in k_list:
            rc._get_freq_shift_exp.cache_clear()
            audio, met = rc._process_channel_dsp_stateless(iq_q, fs, _cfg("nbfm", offs[k]))
            out[f"nbfm{chunk}_k{k}_audio"] = audio
            out[f"nbfm{chunk}_k{k}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
    out["nbfm_k"] = np.array(k_list)
    out["nbfm_args"] = np.array([fs, n, 500], dtype=np.int64)
    # WBFM: config-1 content, defaults (de-emphasis 75us + MPX 15 kHz), offset 0 and +200 kHz
    for ci, off in enumerate([0.0, 200000.0]):
        iq = S.fm_tone_c64(n, fs, seed=510 + ci, carrier_hz=off)
        cfg = rc.ChannelConfig(id="w", capture_id="c", mode="wbfm", offset_hz=off)
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(iq, fs, cfg)
        out[f"wbfm{ci}_sha"] = np.array(S.sha256(iq))
        out[f"wbfm{ci}_audio"] = audio
        out[f"wbfm{ci}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
        out[f"wbfm{ci}_args"] = np.array([fs, n, 510 + ci, int(off)], dtype=np.int64)
    # AM / SSB / filtered NBFM at rates where the reference's order-5 ba-form Butterworths are
    # well conditioned (at >= 96 kS/s the 100 Hz high-pass amplifies 1-ulp input noise to 1e-3,
    # see DESIGN.md): sensitivity of every case below to +-1 ulp input noise is < 5e-7.
    def run(tag, iq, fs_, cfg):
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(iq, fs_, cfg)
        assert audio is not None, tag
        out[f"{tag}_sha"] = np.array(S.sha256(iq))
        out[f"{tag}_audio"] = audio
        out[f"{tag}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])

    iq = S.am_tone_c64(4800, 48000, seed=520, carrier_hz=4800.0, depth=0.8)
    run("am48", iq, 48000, rc.ChannelConfig(id="a", capture_id="c", mode="am", offset_hz=4800.0, enable_agc=True))
    run("am48to24", iq, 48000, rc.ChannelConfig(id="a", capture_id="c", mode="am", offset_hz=4800.0, enable_agc=True,
                                               audio_rate=24000))
    run("am48noagc", iq, 48000, rc.ChannelConfig(id="a", capture_id="c", mode="am", offset_hz=4800.0,
                                                enable_agc=False))
    iq = S.am_tone_c64(9600, 96000, seed=521, carrier_hz=9600.0, depth=0.8)
    run("am96nohp", iq, 96000, rc.ChannelConfig(id="a", capture_id="c", mode="am", offset_hz=9600.0, enable_agc=True,
                                               enable_am_highpass=False))
    for mode, car in (("usb", 2500.0), ("lsb", 4200.0)):   # carriers chosen so the audio passes validate_audio_samples
        iq = S.am_tone_c64(3200, 32000, seed=530, carrier_hz=car, depth=0.8)
        run(f"ssb32{mode}", iq, 32000, rc.ChannelConfig(id="s", capture_id="c", mode="ssb", offset_hz=3200.0,
                                                        enable_agc=True, ssb_mode=mode))
    iq = S.fm_tone_c64(9600, 96000, seed=540, audio_hz=700.0, deviation=4000.0, carrier_hz=12000.0, noise_amp=0.01)
    run("nbfmf96", iq, 96000, rc.ChannelConfig(id="f", capture_id="c", mode="nbfm", offset_hz=12000.0,
                                              enable_deemphasis=True, enable_mpx_filter=False, enable_fm_highpass=True,
                                              fm_highpass_hz=300, enable_fm_lowpass=True, fm_lowpass_hz=3000,
                                              notch_frequencies=[1000.0]))
    # N1: Channel.update_signal_metrics on config-2 content (10 calls -> the SNR branch runs on the 10th)
    iq = S.nbfm_bank_c64(n, fs, seed=500, start=0)
    offs_all = S.nbfm_bank_offsets()
    mets = []
    for k in (0, 13, 31):
        chn = rc.Channel(cfg=_cfg("nbfm", offs_all[k]))
        chn.state = "running"
        for _ in range(10):
            rc._get_freq_shift_exp.cache_clear()
            chn.update_signal_metrics(iq, fs)
        mets.append([chn.rssi_db, chn.snr_db])
    out["sigmet"] = np.array(mets)
    save("chain_analog", **out)


def gen_rawdig():
    """capture.py:415-430: the "raw" and digital-voice branches of _process_channel_dsp_stateless."""
    out = {}
    fs, n = 2400000, 4096
    iq = S.nbfm_bank_c64(n, fs, seed=550)
    out["sha"] = np.array(S.sha256(iq))
    out["args"] = np.array([fs, n, 550], dtype=np.int64)
    for ci, off in enumerate([0.0, 175000.0]):
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(iq, fs, rc.ChannelConfig(id="r", capture_id="c", mode="raw", offset_hz=off))
        out[f"raw{ci}_audio"] = audio
        out[f"raw{ci}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(iq, fs, rc.ChannelConfig(id="d", capture_id="c", mode="p25", offset_hz=off))
        assert audio is None
        out[f"dig{ci}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
    out["offs"] = np.array([0.0, 175000.0])
    # raw IQ above the audio clip limit fails validate_audio_samples: (None, {rssi_db})
    audio, met = rc._process_channel_dsp_stateless((iq * 40).astype(np.complex64), fs,
                                                   rc.ChannelConfig(id="r", capture_id="c", mode="raw", offset_hz=0.0))
    assert audio is None and "signal_power_db" not in met
    out["loud_rssi"] = np.array(met["rssi_db"])
    save("chain_rawdig", **out)


def gen_sam():
    """A14: "sam" mode of _process_channel_dsp_stateless (capture.py:386-399 -> dsp/sam.py:132-269), at rates
    where the order-5 ba-form Butterworths are well conditioned (see gen_chain)."""
    out = {}
    cases = [
        # tag, fs, n, seed, carrier_hz (signal), offset_hz (channel), kwargs
        ("dsb48", 48000, 4800, 560, 4815.0, 4800.0, dict(enable_agc=True)),                    # 15 Hz residual carrier
        ("usb48", 48000, 4800, 561, 4790.0, 4800.0, dict(enable_agc=True, sam_sideband="usb")),
        ("lsb32", 32000, 3200, 562, 3200.0, 3200.0, dict(enable_agc=False, sam_sideband="lsb", audio_rate=16000)),
        ("dsb96", 96000, 9600, 563, 30.0, 0.0, dict(enable_agc=True, enable_am_highpass=False, sam_pll_bandwidth_hz=100.0)),
    ]
    for tag, fs, n, seed, car, off, kw in cases:
        iq = S.am_tone_c64(n, fs, seed=seed, carrier_hz=car, depth=0.7)
        cfg = rc.ChannelConfig(id="s", capture_id="c", mode="sam", offset_hz=off, **kw)
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(iq, fs, cfg)
        assert audio is not None, tag
        out[f"{tag}_sha"] = np.array(S.sha256(iq))
        out[f"{tag}_audio"] = audio
        out[f"{tag}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
        out[f"{tag}_args"] = np.array([fs, n, seed, car, off], dtype=np.float64)
        out[f"{tag}_kw"] = np.array(repr(kw))
    out["tags"] = np.array([c[0] for c in cases])
    save("chain_sam", **out)


def gen_nr():
    """A14: spectral_noise_reduction (dsp/filters.py:346-460) as reached through the nbfm / wbfm branches of
    _process_channel_dsp_stateless with enable_noise_reduction=True."""
    out = {}
    offs = S.nbfm_bank_offsets()
    cases = [
        # tag, mode, fs, n, seed, channel offset, extra cfg
        ("nbfm24", "nbfm", 2400000, 120000, 570, offs[13], dict()),                 # 233 frames, row shrinks to 119808
        ("nbfm24_18db", "nbfm", 2400000, 60000, 571, offs[20], dict(noise_reduction_db=18.0)),
        ("wbfm24", "wbfm", 2400000, 60000, 572, 200000.0, dict()),                  # de-emphasis + MPX, then NR
        ("nbfm48", "nbfm", 48000, 4800, 573, 6000.0, dict()),                       # no resampling: audio = 4608 samples
        ("short", "nbfm", 48000, 1000, 574, 6000.0, dict()),                        # < 1024 samples: passes through
    ]
    for tag, mode, fs, n, seed, off, kw in cases:
        if fs == 2400000 and mode == "nbfm":
            iq = S.nbfm_bank_c64(n, fs, seed=seed)
        elif mode == "wbfm":
            iq = S.fm_tone_c64(n, fs, seed=seed, carrier_hz=off, noise_amp=0.2)
        else:
            iq = S.fm_tone_c64(n, fs, seed=seed, audio_hz=700.0, deviation=3000.0, carrier_hz=off, noise_amp=0.2)
        cfg = _cfg(mode, off) if mode == "nbfm" else rc.ChannelConfig(id="w", capture_id="c", mode="wbfm", offset_hz=off)
        cfg.enable_noise_reduction = True
        for k, v in kw.items():
            setattr(cfg, k, v)
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(iq, fs, cfg)
        assert audio is not None, tag
        out[f"{tag}_sha"] = np.array(S.sha256(iq))
        out[f"{tag}_audio"] = audio
        out[f"{tag}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
        out[f"{tag}_args"] = np.array([fs, n, seed, off, kw.get("noise_reduction_db", 12.0)], dtype=np.float64)
        out[f"{tag}_mode"] = np.array(mode)
    out["tags"] = np.array([c[0] for c in cases])
    save("chain_nr", **out)


def gen_a7():
    out = {}
    cases = [(1_000_000, 25000, 40, 701), (8_000_000, 25000, 320, 702), (10_000_000, 9765, 1024, 703)]
    for ci, (fs, bw, M, seed) in enumerate(cases):
        ch = PolyphaseChannelizer(fs, bw)
        assert ch.channel_count == M
        n1 = M * 12 + M // 2 + 37      # not a multiple of the hop: trailing samples dropped
        n2 = M * 7 + 5
        x = S.noise_c64(n1 + n2, seed)
        r1 = np.array(ch.process(x[:n1]))
        r2 = np.array(ch.process(x[n1:]))
        out[f"c{ci}_args"] = np.array([fs, bw, M, seed, n1, n2], dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(x))
        out[f"c{ci}_r1"] = r1.astype(np.complex64)
        out[f"c{ci}_r2"] = r2.astype(np.complex64)
        out[f"c{ci}_hist"] = ch.arm_history.copy()
        if ci == 0:
            out["c0_arms"] = ch.arms.copy()
        out[f"c{ci}_arms_sha"] = np.array(S.sha256(ch.arms))
        out[f"c{ci}_ex5"] = ch.extract_channel(list(r1), 5)
    # short input (< M) returns no hops
    ch = PolyphaseChannelizer(1_000_000, 25000)
    out["short_len"] = np.array(len(ch.process(S.noise_c64(39, 1))))
    out["n_cases"] = np.array(len(cases))
    save("a7_pfb", **out)


def gen_a8():
    out = {}
    fs = 2400000
    for ci, N in enumerate([512, 1024, 2048, 4096]):
        iq = S.fm_tone_c64(N + 100, fs, seed=800 + ci, deviation=20000.0, carrier_hz=123456.0, noise_amp=0.01)
        r = ScipyFFTBackend(N).execute(iq, fs)
        out[f"c{ci}_args"] = np.array([N, fs, 800 + ci], dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(iq))
        out[f"c{ci}_power"] = r.power_db
        out[f"c{ci}_freqs"] = r.freqs
        out[f"c{ci}_bin"] = np.array(r.bin_hz)
    r = ScipyFFTBackend(1024).execute(S.noise_c64(100, 1), fs)
    out["short_power"] = r.power_db
    out["n_cases"] = np.array(4)
    save("a8_spectrum", **out)


def gen_c4fm():
    """A9-A11: full C4FMDemodulator.demodulate on streamed calls."""
    out = {}
    cases = [
        # (fs, n_total, call_len, seed, snr_db, foff_hz, silence)
        (48000, 48000 * 3, 4800, 1000, 20.0, 120.0, None),
        (48000, 48000 * 2, 4800, 1001, 8.0, -300.0, None),
        (50000, 50000 * 2, 5000, 1002, 30.0, 250.0, None),
        (19200, 19200 * 3, 1920, 1003, 15.0, -80.0, None),
        (48000, 48000 * 4, 4800, 1004, 20.0, 0.0, (60000, 110000)),   # fine-sync loss + buffer shifts
        (48000, 48000 * 1, 4800, 1005, -30.0, 0.0, None),             # essentially noise
        (48000, 48000 * 2, 7777, 1006, 25.0, 390.0, None),            # ragged call length
    ]
    for ci, (fs, n, call, seed, snr, foff, sil) in enumerate(cases):
        iq, _ = S.c4fm_iq(n, fs, seed, snr_db=snr, freq_offset_hz=foff, silence=sil)
        d = rc4.C4FMDemodulator(sample_rate=fs)
        dib, soft, counts = [], [], []
        for s in range(0, n, call):
            a, b = d.demodulate(iq[s:s + call])
            dib.append(a)
            soft.append(b)
            counts.append(len(a))
        out[f"c{ci}_args"] = np.array([fs, n, call, seed, int(round(snr * 10)), int(round(foff * 10)),
                                       -1 if sil is None else sil[0], -1 if sil is None else sil[1]],
                                      dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(iq))
        out[f"c{ci}_dibits"] = np.concatenate(dib).astype(np.uint8)
        out[f"c{ci}_soft"] = np.concatenate(soft).astype(np.float32)
        out[f"c{ci}_counts"] = np.array(counts, dtype=np.int32)
        out[f"c{ci}_state"] = np.array([d._sync_count, int(d._fine_sync), d._equalizer.pll,
                                        d._equalizer.gain, float(d._sample_point)])
        print(f"  c4fm case {ci}: syms={sum(counts)} syncs={d._sync_count} fine={d._fine_sync} "
              f"pll={d._equalizer.pll:.4f} gain={d._equalizer.gain:.4f}")
    out["n_cases"] = np.array(len(cases))
    # filter designs (host-side, but pinned so the product's taps are the reference's taps)
    for fs in (48000, 50000, 19200):
        d = rc4.C4FMDemodulator(sample_rate=fs)
        out[f"lpf_{fs}"] = d._baseband_lpf
        out[f"rrc_{fs}"] = d._rrc_filter
    save("c4fm", **out)
    # The 129x8 MMSE interpolator coefficient table (GNU Radio / SDRTrunk constants) is data the
    # algorithm is defined by; stored as binary fixtures for the oracle and for the product.
    taps = np.ascontiguousarray(rc4._interpolator.TAPS, dtype=np.float32)
    np.save(os.path.join(GOLD, "mmse_interp_taps_f32.npy"), taps)
    os.makedirs(os.path.join(REPO, "wavecap-sdr_amd", "wavehip", "data"), exist_ok=True)
    np.save(os.path.join(REPO, "wavecap-sdr_amd", "wavehip", "data", "mmse_interp_taps_f32.npy"), taps)


def gen_trunk():
    """A4/A5 (control-channel IQ path of trunking/system.py:1735-1779) and A13 (cc_scanner)."""
    from scipy import signal as ss
    from wavecapsdr.dsp.filters import fir_decimate
    from wavecapsdr.trunking.cc_scanner import ControlChannelScanner

    out = {}
    fs, d1, d2 = 2_400_000, 10, 5
    t1 = ss.firwin(157, 0.8 / d1, window=("kaiser", 7.857))
    t2 = ss.firwin(73, 0.8 / d2, window=("kaiser", 7.857))
    z1t = ss.lfilter_zi(t1, 1.0).astype(np.complex128)
    z2t = ss.lfilter_zi(t2, 1.0).astype(np.complex128)
    # the NCO lives in a closure (system.py:1434-1466); its arithmetic is restated line by line here
    state = dict(idx=0, last=0.0, z1=None, z2=None)

    def nco(iq, off):
        if off == 0.0 or iq.size == 0:
            return iq
        if off != state["last"]:
            state["idx"] = 0
            state["last"] = off
        n = np.arange(iq.size, dtype=np.float64) + state["idx"]
        phase = -2.0 * np.pi * off * n / fs
        shift = np.exp(1j * phase).astype(np.complex64)
        y = np.asarray(iq.astype(np.complex64, copy=False) * shift, dtype=np.complex64)
        state["idx"] += iq.size
        if state["idx"] >= fs:
            state["idx"] %= fs
        return y

    lens = [120000, 120000, 99991, 120000, 50, 120000]     # ragged + a tiny call; > 1 s in total wraps the index
    offs = [312500.0, 312500.0, 312500.0, -100000.0, -100000.0, -100000.0]
    x = S.c4fm_iq(sum(lens), fs, 1200, snr_db=25.0, freq_offset_hz=312500.0)[0]
    pos = 0
    outs = []
    for n, off in zip(lens, offs):
        c = nco(x[pos:pos + n], off)
        pos += n
        if state["z1"] is None:
            state["z1"] = z1t * c[0]
        y1, state["z1"] = fir_decimate(c, t1, d1, zi=state["z1"])
        if state["z2"] is None:
            state["z2"] = z2t * y1[0]
        y2, state["z2"] = fir_decimate(y1, t2, d2, zi=state["z2"])
        outs.append(np.asarray(y2, dtype=np.complex64))
    out["ddc_args"] = np.array([fs, d1, d2, 1200], dtype=np.int64)
    out["ddc_lens"] = np.array(lens, dtype=np.int64)
    out["ddc_offs"] = np.array(offs)
    out["ddc_sha"] = np.array(S.sha256(x))
    out["ddc_out"] = np.concatenate(outs)
    out["ddc_counts"] = np.array([len(o) for o in outs], dtype=np.int64)
    # scanner: 3 carriers in a 2.4 MS/s buffer
    n = 240000
    center = 415.0e6
    chans = [center + 300e3, center - 450e3, center + 812.5e3, center + 50e3]
    t = np.arange(n) / fs
    rng = np.random.default_rng(77)
    w = (0.2 * np.exp(2j * np.pi * 300e3 * t) + 0.05 * np.exp(2j * np.pi * -450e3 * t)
         + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    sc = ControlChannelScanner(center_hz=center, sample_rate=fs, control_channels=chans, sync_check_enabled=False)
    meas = [sc._measure_channel(w, f) for f in chans]
    out["scan_sha"] = np.array(S.sha256(w))
    out["scan_args"] = np.array([fs, n, 77], dtype=np.int64)
    out["scan_offsets"] = np.array([f - center for f in chans])
    out["scan_meas"] = np.array([[m.power_db, m.peak_power_db, m.noise_floor_db, m.snr_db, m.sample_count]
                                 for m in meas])
    # sync-pattern check (cc_scanner.py:266-353): a real C4FM control channel at +300 kHz, a CW at -450 kHz
    w2 = (S.c4fm_iq(n, fs, 1300, snr_db=30.0, freq_offset_hz=300e3, amp=0.2)[0]
          + (0.05 * np.exp(2j * np.pi * -450e3 * t)).astype(np.complex64)).astype(np.complex64)
    sc2 = ControlChannelScanner(center_hz=center, sample_rate=fs, control_channels=chans, sync_check_enabled=True)
    meas2 = [sc2._measure_channel(w2, f) for f in chans]
    out["scan2_sha"] = np.array(S.sha256(w2))
    out["scan2_meas"] = np.array([[m.power_db, m.peak_power_db, m.noise_floor_db, m.snr_db, float(m.sync_detected)]
                                  for m in meas2])
    print("  scanner sync flags:", [m.sync_detected for m in meas2], [round(m.snr_db, 1) for m in meas2])
    save("trunk", **out)


def gen_recorder():
    """N3: VoiceRecorder.process_iq front-end (trunking/system.py:453-656) for two recorders on the same 6 MS/s
    buffer.  The decimated IQ is a local of process_iq, so scipy.signal.lfilter is wrapped while it runs and the
    stage outputs are taken from the wrapper (stage 2 output [::D2] = decimated_iq)."""
    from scipy import signal as ss
    from wavecapsdr.trunking import system as rsys

    fs, n_chunk = 6_000_000, 150_000
    lens = [n_chunk, n_chunk, 149_987, 77, n_chunk]
    x = S.c4fm_iq(sum(lens), fs, 1210, snr_db=25.0, freq_offset_hz=412_500.0)[0]
    x = (x + np.roll(x, 7) * np.exp(-2j * np.pi * 1_237_500.0 * np.arange(x.size) / fs)).astype(np.complex64)
    recs = []
    for rid, off in (("r0", 412_500.0), ("r1", -825_000.0)):
        r = rsys.VoiceRecorder(id=rid, system_id="s")
        r.setup_decimation_filter(fs, 48000)
        r.state = "recording"
        r._voice_channel = object()       # only tested for None before the DSP front-end
        r.offset_hz = off
        recs.append(r)
    out = {"args": np.array([fs, 1210], dtype=np.int64), "lens": np.array(lens, dtype=np.int64),
           "sha": np.array(S.sha256(x)), "offsets": np.array([r.offset_hz for r in recs]),
           "factors": np.array([recs[0]._stage1_decim_factor, recs[0]._stage2_decim_factor], dtype=np.int64),
           "t1": recs[0]._stage1_filter_taps, "t2": recs[0]._stage2_filter_taps}
    real = ss.lfilter
    got = {0: [], 1: []}
    for ri, r in enumerate(recs):
        pos = 0
        for ci, ln in enumerate(lens):
            if ri == 1 and ci == 1:       # recorder 1 idle during chunk 1: skipped, state kept
                pos += ln
                continue
            calls = []

            def spy(b, a, xx, axis=-1, zi=None):
                res = real(b, a, xx, axis=axis, zi=zi)
                calls.append(res[0] if zi is not None else res)
                return res

            ss.lfilter = spy
            try:
                try:
                    r.process_iq(x[pos:pos + ln], fs)
                except Exception:          # the vocoder hand-off after the front-end has no event loop here
                    pass
            finally:
                ss.lfilter = real
            pos += ln
            assert len(calls) >= 2, (ri, ci, len(calls))
            got[ri].append(np.asarray(calls[1][::r._stage2_decim_factor]).astype(np.complex64))
    for ri in (0, 1):
        out[f"r{ri}_out"] = np.concatenate(got[ri])
        out[f"r{ri}_counts"] = np.array([len(v) for v in got[ri]], dtype=np.int64)
    save("trunk_recorder", **out)


def gen_classifier():
    """N4: ChannelClassifier.update / classify (channel_classifier.py:62-238)."""
    from wavecapsdr.channel_classifier import ChannelClassifier

    p, freqs = S.classifier_frames()
    cl = ChannelClassifier(min_collection_seconds=0.0)
    for f in range(p.shape[0]):
        cl.update(p[f].tolist(), freqs.tolist(), 851_000_000.0, 2_400_000.0)      # capture.py:2395-2404 passes lists
    st = np.array([[s.sum, s.sum_sq, s.count, s.min_val, s.max_val] for _, s in sorted(cl._bin_stats.items())])
    res = cl.classify(force=True)
    kinds = {"control": 0, "voice": 1, "variable": 2, "unknown": 3}
    save("classifier", sha=np.array(S.sha256(p)), stats=st,
         chans=np.array([[c.freq_hz, c.power_db, c.std_dev_db, kinds[c.channel_type]] for c in res]),
         sample_count=np.array(cl.sample_count))


def gen_nid():
    """N2: bch_decode (dsp/fec/bch.py:533-658) on random words, and the NID events of
    P25P1MessageFramer.process_batch (decoders/p25_framer.py:471-617) on a synthetic dibit stream."""
    from wavecapsdr.dsp.fec.bch import bch_decode
    from wavecapsdr.decoders import p25_framer as rfr

    rng = np.random.default_rng(1410)
    words, tracked, data, errs = [], [], [], []
    for k in range(360):
        d16 = int(rng.integers(0, 1 << 16))
        cw = S.bch_encode(d16)
        n_err = int(rng.integers(0, 15)) if k % 6 else int(rng.integers(10, 13))
        w = cw
        for f in rng.choice(63, size=n_err, replace=False):
            w ^= 1 << int(f)
        if k % 5 == 0:
            w = int(rng.integers(0, 1 << 63))                       # arbitrary word
        tn = [0, (d16 >> 4) & 0xFFF, int(rng.integers(1, 0x1000))][k % 3]
        bits = np.array([(w >> (62 - i)) & 1 for i in range(63)], dtype=np.uint8)
        dd, ee = bch_decode(bits, tn if tn else None)
        words.append(w); tracked.append(tn); data.append(dd); errs.append(ee)
    out = dict(words=np.array(words, dtype=np.uint64), tracked=np.array(tracked, dtype=np.int32),
               data=np.array(data, dtype=np.int32), errors=np.array(errs, dtype=np.int32))
    # framer
    dib, soft, heads = S.nid_stream()
    fr = rfr.P25P1MessageFramer()
    fr.start()
    log = []
    real = rfr.bch_decode

    def spy(cw, tn=None):
        r = real(cw, tn)
        log.append((fr._debug_symbol_count - 1, int(r[0]), int(r[1]), int(tn or 0)))
        return r

    rfr.bch_decode = spy
    try:
        lens = [137, 61, 300, 5, 211]
        pos, k = 0, 0
        counts = []
        while pos < dib.size:
            ln = lens[k % len(lens)]
            k += 1
            try:
                counts.append(fr.process_batch(soft[pos:pos + ln], dib[pos:pos + ln]))
            except Exception as e:        # message assembly downstream of the NID is not part of this row
                counts.append(-1)
            pos += ln
    finally:
        rfr.bch_decode = real
    out.update(sha=np.array(S.sha256(dib) + S.sha256(soft)), lens=np.array(lens, dtype=np.int64),
               attempts=np.array(log, dtype=np.int64),                       # (index, data, errors, tracked) per BCH call
               events=np.array([(i, (d >> 4) & 0xFFF, d & 0xF, e) for i, d, e, _ in log if e >= 0], dtype=np.int64),
               nid_counts=np.array(counts, dtype=np.int64))
    save("nid", **out)


def gen_status():
    """N2: P25Decoder._strip_status_symbols (decoders/p25.py:2796-2862) on random dibit rows: lengths around the
    36-dibit period, the TSDU length (101+), long rows; initial counters 0, 21 (TSDU), 35 (first dibit is a status
    symbol), 36 / 40 (the counter never meets 36: nothing stripped), -3."""
    from wavecapsdr.decoders.p25 import P25Decoder

    dec = P25Decoder.__new__(P25Decoder)      # the method reads class state only
    rng = np.random.default_rng(2796)
    out, k = {}, 0
    for n in (0, 1, 13, 14, 15, 35, 36, 37, 72, 101, 120, 360, 2001):
        for c0 in (0, 21, 35, 36, 40, -3):
            d = rng.integers(0, 4, size=n).astype(np.uint8)
            r = dec._strip_status_symbols(d, initial_counter=c0)
            out[f"c{k}_in"], out[f"c{k}_c0"], out[f"c{k}_out"] = d, np.array(c0), np.asarray(r, dtype=np.uint8)
            k += 1
    out["n_cases"] = np.array(k)
    save("status", **out)


def gen_chain4():
    """Config 4, chained: IQ -> C4FMDemodulator.demodulate (dsp/p25/c4fm.py:2528) -> P25P1MessageFramer.process_batch
    (decoders/p25_framer.py:471-617), fed in 100 ms calls as cli.py:700-716 / decoders/p25.py:1961 do: every BCH
    attempt and every NID event of the reference on a carrier that transmits frame heads."""
    from wavecapsdr.decoders import p25_framer as rfr

    fs, call = 48000, 4800
    out = {}
    for ci, (seed, snr, foff) in enumerate(((1500, 20.0, 150.0), (1520, 12.0, -310.0))):
        iq, tx = S.p25_head_stream_iq(fs, seed, 3, snr, foff)
        dem = rc4.C4FMDemodulator(sample_rate=fs)
        fr = rfr.P25P1MessageFramer()
        fr.start()
        log = []
        real = rfr.bch_decode

        def spy(cw, tn=None):
            r = real(cw, tn)
            log.append((fr._debug_symbol_count - 1, int(r[0]), int(r[1]), int(tn or 0)))
            return r

        rfr.bch_decode = spy
        counts, dibs, softs = [], [], []
        try:
            for s0 in range(0, len(iq), call):
                d, sf = dem.demodulate(iq[s0:s0 + call])
                counts.append(len(d)); dibs.append(np.asarray(d, dtype=np.uint8)); softs.append(np.asarray(sf, dtype=np.float32))
                if len(d):
                    try:
                        fr.process_batch(np.asarray(sf, dtype=np.float32), np.asarray(d, dtype=np.uint8))
                    except Exception:      # message assembly downstream of the NID is not part of this row
                        pass
        finally:
            rfr.bch_decode = real
        ev = np.array([(i, (d >> 4) & 0xFFF, d & 0xF, e) for i, d, e, _ in log if e >= 0], dtype=np.int64).reshape(-1, 4)
        out[f"c{ci}_args"] = np.array([fs, call, seed, int(snr * 10), int(foff * 10), 3], dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(iq))
        out[f"c{ci}_counts"] = np.array(counts, dtype=np.int32)
        out[f"c{ci}_dibits"] = np.concatenate(dibs)
        out[f"c{ci}_soft"] = np.concatenate(softs)
        out[f"c{ci}_attempts"] = np.array(log, dtype=np.int64).reshape(-1, 4)
        out[f"c{ci}_events"] = ev
        print(f"chain4 case {ci}: {len(iq)} samples, {sum(counts)} dibits, {len(log)} BCH attempts, {len(ev)} NID events")
    out["n_cases"] = np.array(2)
    save("chain4", **out)


def gen_c4fm_big():
    """A9-A11 at the CALLER'S call sizes: the control-channel monitor hands demodulate() max(20 000, 1.5 fs) samples
    (trunking/system.py:1548-1549 -> control_channel.py:230-231), cli.py:755 a whole recording.  The reference's block
    processing is not cut-invariant (symbol extraction of the whole call uses the call-start equaliser; sync search runs
    on the FINAL 65 536-sample phase buffer, symbols shifted out of it get index -1, c4fm.py:704-728), so these pin the
    one-call results: single 72 000 / 75 000-sample calls, three consecutive 72 000-sample calls (carried state, a buffer
    shift inside a call), one 200 000-sample call (>= 2 shifts: early symbols index -1), one 150 000-sample call at sps 4."""
    out = {}
    cases = [
        # (fs, seed, snr_db, foff_hz, [call lengths])
        (48000, 1100, 20.0, 140.0, [72000]),
        (50000, 1101, 18.0, -220.0, [75000]),
        (48000, 1102, 22.0, 90.0, [72000, 72000, 72000]),
        (48000, 1103, 20.0, -60.0, [200000]),
        (19200, 1104, 16.0, 40.0, [150000, 28800]),
        (48000, 1105, 25.0, 310.0, [4800, 100000, 333, 72000]),
    ]
    for ci, (fs, seed, snr, foff, calls) in enumerate(cases):
        n = sum(calls)
        iq, _ = S.c4fm_iq(n, fs, seed, snr_db=snr, freq_offset_hz=foff)
        d = rc4.C4FMDemodulator(sample_rate=fs)
        dib, soft, counts, pos = [], [], [], 0
        for m in calls:
            a, b = d.demodulate(iq[pos:pos + m])
            pos += m
            dib.append(a); soft.append(b); counts.append(len(a))
        out[f"c{ci}_args"] = np.array([fs, seed, int(round(snr * 10)), int(round(foff * 10))], dtype=np.int64)
        out[f"c{ci}_calls"] = np.array(calls, dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(iq))
        out[f"c{ci}_dibits"] = np.concatenate(dib).astype(np.uint8)
        out[f"c{ci}_soft"] = np.concatenate(soft).astype(np.float32)
        out[f"c{ci}_counts"] = np.array(counts, dtype=np.int32)
        out[f"c{ci}_state"] = np.array([d._sync_count, int(d._fine_sync), d._equalizer.pll, d._equalizer.gain,
                                        float(d._sample_point), d._buffer_pointer, d._symbols_since_sync])
        print(f"  c4fm_big case {ci}: syms={sum(counts)} syncs={d._sync_count} fine={d._fine_sync} "
              f"bp={d._buffer_pointer} pll={d._equalizer.pll:.4f} gain={d._equalizer.gain:.4f}")
    out["n_cases"] = np.array(len(cases))
    # muted input (exact zeros through both FIRs): the discriminator's products are signed zeros, its phase 0 or pi by
    # their signs (np.arctan2), 100 ms calls and one 72 000-sample call
    iq, _ = S.c4fm_muted_iq()
    for tag, calls in (("muted_a", [4800] * 20), ("muted_b", [72000, 24000])):
        d = rc4.C4FMDemodulator(sample_rate=48000)
        dib, soft, counts, pos = [], [], [], 0
        for m in calls:
            a, b = d.demodulate(iq[pos:pos + m])
            pos += m
            dib.append(a); soft.append(b); counts.append(len(a))
        out[f"{tag}_calls"] = np.array(calls, dtype=np.int64)
        out[f"{tag}_dibits"] = np.concatenate(dib).astype(np.uint8)
        out[f"{tag}_soft"] = np.concatenate(soft).astype(np.float32)
        out[f"{tag}_counts"] = np.array(counts, dtype=np.int32)
    out["muted_sha"] = np.array(S.sha256(iq))
    save("c4fm_big", **out)


def gen_cqpsk_big():
    """A12 Phase-2 chain at large / odd call shapes: demodulate() re-seeds the matched filter with zi = state * iq[0]
    on every call (cqpsk.py:283-285), so results depend on where calls are cut: one 140 000-sample call; and
    samples_per_symbol = 5 and 5.2083 (sps / 2 not an integer), CQPSKDemodulator and MuellerMullerTED."""
    from wavecapsdr.dsp.p25.cqpsk import CostasLoop, CQPSKDemodulator as RefCQPSK
    from wavecapsdr.dsp.p25.symbol_timing import MuellerMullerTED

    out = {}
    cases = [(48000, 12000, 1510, 24.0, 35.0, [140000]), (60000, 12000, 1511, 22.0, -50.0, [2500, 9000, 21]),
             (62500, 12000, 1512, 26.0, 20.0, [2500, 7000])]
    for ci, (fs, sr, seed, snr, foff, calls) in enumerate(cases):
        n = sum(calls)
        iq, _ = S.dqpsk_iq(n, fs, seed, symbol_rate=sr, snr_db=snr, freq_offset_hz=foff)
        d = RefCQPSK(sample_rate=fs, symbol_rate=sr)
        dib, pos = [], 0
        for m in calls:
            dib.append(d.demodulate(iq[pos:pos + m]))
            pos += m
        out[f"c{ci}_args"] = np.array([fs, sr, seed, int(snr * 10), int(foff * 10)], dtype=np.int64)
        out[f"c{ci}_calls"] = np.array(calls, dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(iq))
        out[f"c{ci}_dibits"] = np.concatenate(dib).astype(np.uint8)
        out[f"c{ci}_counts"] = np.array([len(x) for x in dib], dtype=np.int64)
        out[f"c{ci}_state"] = np.array([d._carrier_loop._phase, d._carrier_loop._freq, d._timing_recovery._phase,
                                        d._timing_recovery._integrator])
        print(f"  cqpsk_big case {ci}: dibits={sum(len(x) for x in dib)}")
    out["n_cases"] = np.array(len(cases))
    # a muted stretch (exact zeros): the rotated value of a zero sample is (+-0, +-0), its angle 0 or +-pi -> detector error 0
    iq, _ = S.dqpsk_muted_iq()
    d = RefCQPSK(sample_rate=48000, symbol_rate=12000)
    parts = [d.demodulate(iq[:13000]), d.demodulate(iq[13000:])]
    out["muted_sha"] = np.array(S.sha256(iq))
    out["muted_dibits"] = np.concatenate(parts).astype(np.uint8)
    out["muted_counts"] = np.array([len(x) for x in parts], dtype=np.int64)
    # MuellerMullerTED standalone at sps 5 and 5.2083 on a Costas-corrected carrier, two calls each
    iq, _ = S.dqpsk_iq(6000, 60000, 1513, symbol_rate=12000, snr_db=22.0, freq_offset_hz=30.0)
    x = CostasLoop().process_block(iq.astype(np.complex128))
    out["mm_sha"] = np.array(S.sha256(iq))
    out["mm_in"] = x
    for tag, sps in (("s5", 5.0), ("s52", 62500 / 12000)):
        mm = MuellerMullerTED(sps)
        a, b = mm.process_block(x[:2500]), mm.process_block(x[2500:])
        out[f"mm_{tag}_sps"] = np.array([sps])
        out[f"mm_{tag}_counts"] = np.array([len(a[0]), len(b[0])], dtype=np.int64)
        out[f"mm_{tag}_sym"] = np.concatenate([a[0], b[0]])
        out[f"mm_{tag}_dec"] = np.concatenate([a[1], b[1]])
        out[f"mm_{tag}_err"] = np.concatenate([a[2], b[2]])
    save("cqpsk_big", **out)


def gen_caprate():
    """A14 at CAPTURE rates (250 kS/s, 2.4 MS/s), where the live loop runs AM / SSB / SAM channels
    (capture.py:298-439 on chunks of max(8192, fs // 20) samples).  The reference's order-5 ba-form Butterworths are
    ill-conditioned there: what IS reproducible is the OUTCOME CLASS the live loop acts on -- audio or
    None (validate_audio_samples, validation.py:41-52 -> capture.py:323-325, 2593-2595) -- and the metrics.  Per case:
    class, rssi_db / signal_power_db, the audio, and the measured sensitivity of that audio to a +-1-ulp perturbation of
    the input (`sens`, peak-relative; the class under the perturbation must not change or the case is dropped): the
    device must reproduce class and metrics, and the audio to max(1e-5, 4 * sens) wherever sens < 1e-3."""
    out = {}
    cases = []
    for fs in (250_000, 2_400_000):
        n = max(8192, fs // 20)
        car = 30_000.0
        cases += [
            (f"am{fs // 1000}", fs, n, 580, "am", car, car, dict()),
            (f"am{fs // 1000}_nohp", fs, n, 581, "am", car, car, dict(enable_am_highpass=False)),
            (f"am{fs // 1000}_noagc", fs, n, 582, "am", car, car, dict(enable_agc=False)),
            (f"usb{fs // 1000}", fs, n, 583, "ssb", car + 1200.0, car, dict(ssb_mode="usb")),
            (f"lsb{fs // 1000}", fs, n, 584, "ssb", car - 900.0, car, dict(ssb_mode="lsb")),
            (f"sam{fs // 1000}", fs, n, 585, "sam", car + 12.0, car, dict()),
            (f"sam{fs // 1000}_usb", fs, n, 586, "sam", car - 8.0, car, dict(sam_sideband="usb")),
        ]
    kept = []
    for tag, fs, n, seed, mode, sig_car, off, kw in cases:
        iq = S.am_tone_c64(n, fs, seed=seed, carrier_hz=sig_car, depth=0.7)
        cfg = rc.ChannelConfig(id="c", capture_id="c", mode=mode, offset_hz=off, **kw)
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(iq, fs, cfg)
        # +-1 ulp on a random half of the float32 components
        rng = np.random.default_rng(seed + 1)
        f = iq.view(np.float32).copy()
        pick = rng.random(f.size) < 0.5
        f[pick] = np.nextafter(f[pick], np.where(rng.random(int(pick.sum())) < 0.5, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))
        rc._get_freq_shift_exp.cache_clear()
        audio2, met2 = rc._process_channel_dsp_stateless(f.view(np.complex64), fs, cfg)
        cls, cls2 = audio is not None, audio2 is not None
        sens = float("nan")
        if cls and cls2:
            sens = float(np.max(np.abs(audio - audio2)) / max(float(np.max(np.abs(audio))), 1e-30))
        print(f"  caprate {tag}: audio={'yes' if cls else 'None'} (perturbed: {'yes' if cls2 else 'None'}) "
              f"rssi={met.get('rssi_db'):.4f} sp={met.get('signal_power_db')} sens={sens:.3g} "
              f"peak={float(np.max(np.abs(audio))) if cls else float('nan'):.4g}")
        if cls != cls2:
            print(f"    -> class not stable under +-1 ulp: dropped")
            continue
        kept.append(tag)
        out[f"{tag}_sha"] = np.array(S.sha256(iq))
        out[f"{tag}_args"] = np.array([fs, n, seed, sig_car, off], dtype=np.float64)
        out[f"{tag}_mode"] = np.array(mode)
        out[f"{tag}_kw"] = np.array(repr(kw))
        out[f"{tag}_class"] = np.array(int(cls))
        out[f"{tag}_met"] = np.array([met["rssi_db"], met.get("signal_power_db", np.nan)])
        out[f"{tag}_met_pert"] = np.array([met2["rssi_db"], met2.get("signal_power_db", np.nan)])
        out[f"{tag}_sens"] = np.array(sens)
        if cls:
            out[f"{tag}_audio"] = audio
    out["tags"] = np.array(kept)
    save("chain_caprate", **out)


def gen_zero():
    """FM chains on input with EXACT zero samples (tests/signals.nbfm_zero_gap_i16): the discriminator value of a sample
    whose conjugate product is exactly zero hangs on IEEE signed zeros in the reference (numpy's complex64 product of a
    zero sample with the mixer phase, then np.angle = arctan2: (+-0, -0) -> +-pi) -- pinned here from the reference itself."""
    fs, n = 2400000, 120000
    i16 = S.nbfm_zero_gap_i16(n, fs)
    z = i16.astype(np.float32) / 32768.0
    z = (z[0::2] + 1j * z[1::2]).astype(np.complex64)
    out = {"sha_i16": np.array(S.sha256(i16)), "args": np.array([fs, n], dtype=np.int64)}
    offs = [S.nbfm_bank_offsets(32)[k] for k in (12, 15)] + [0.0]
    for k, off in enumerate(offs):
        rc._get_freq_shift_exp.cache_clear()
        audio, met = rc._process_channel_dsp_stateless(z, fs, _cfg("nbfm", off))
        out[f"nbfm{k}_off"] = np.array([off])
        out[f"nbfm{k}_audio"] = audio
        out[f"nbfm{k}_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
        fm = rfm.quadrature_demod(z if off == 0.0 else rc.freq_shift(z, off, fs), fs)
        print(f"  zero case nbfm off={off}: discriminator of the muted stretch takes values {np.unique(np.round(fm[30001:30600], 3))[:6]}")
    rc._get_freq_shift_exp.cache_clear()
    audio, met = rc._process_channel_dsp_stateless(z, fs, rc.ChannelConfig(id="w", capture_id="c", mode="wbfm", offset_hz=200000.0))
    out["wbfm_audio"] = audio
    out["wbfm_met"] = np.array([met["rssi_db"], met["signal_power_db"]])
    save("chain_zero", **out)


def gen_blanker():
    """A14: noise_blanker (dsp/filters.py:267-343) on float32 audio with impulses; even and odd lengths."""
    from wavecapsdr.dsp.filters import noise_blanker
    rng = np.random.default_rng(1500)
    out = {}
    for tag, n, db, w in (("even", 6000, 10.0, 3), ("odd", 4001, 8.0, 5), ("w0", 1000, 12.0, 0), ("quiet", 512, 10.0, 3)):
        x = (0.1 * rng.standard_normal(n)).astype(np.float32)
        idx = rng.choice(n, size=max(1, n // 200), replace=False)
        x[idx] += rng.choice([-1.0, 1.0], size=idx.size).astype(np.float32) * rng.uniform(0.5, 3.0, idx.size).astype(np.float32)
        x[0] = 2.0
        x[-1] = -2.0
        if tag == "quiet":
            x[:] = 0
        out[f"{tag}_in"] = x
        out[f"{tag}_out"] = noise_blanker(x.copy(), threshold_db=db, blanking_width=w)
        out[f"{tag}_args"] = np.array([db, w])
    out["tags"] = np.array(["even", "odd", "w0", "quiet"])
    save("blanker", **out)


def gen_cqpsk():
    """A12: Phase-2 CQPSK chain (dsp/p25/cqpsk.py) and the standalone GardnerTED."""
    from wavecapsdr.dsp.p25.cqpsk import CQPSKDemodulator as RefCQPSK
    from wavecapsdr.dsp.p25.symbol_timing import GardnerTED

    out = {}
    cases = [(48000, 12000, 24000, 1501, 25.0, 40.0), (48000, 12000, 24000, 1502, 12.0, -150.0),
             (96000, 12000, 24000, 1503, 30.0, 10.0)]
    for ci, (fs, sr, n, seed, snr, foff) in enumerate(cases):
        iq, _ = S.dqpsk_iq(n, fs, seed, symbol_rate=sr, snr_db=snr, freq_offset_hz=foff)
        d = RefCQPSK(sample_rate=fs, symbol_rate=sr)
        calls = [7000, 33, 9000, n - 16033]
        dib, pos = [], 0
        for m in calls:
            dib.append(d.demodulate(iq[pos:pos + m]))
            pos += m
        out[f"c{ci}_args"] = np.array([fs, sr, n, seed, int(snr * 10), int(foff * 10)], dtype=np.int64)
        out[f"c{ci}_calls"] = np.array(calls, dtype=np.int64)
        out[f"c{ci}_sha"] = np.array(S.sha256(iq))
        out[f"c{ci}_dibits"] = np.concatenate(dib).astype(np.uint8)
        out[f"c{ci}_counts"] = np.array([len(x) for x in dib], dtype=np.int64)
        out[f"c{ci}_state"] = np.array([d._carrier_loop._phase, d._carrier_loop._freq, d._timing_recovery._phase,
                                        d._timing_recovery._integrator])
    out["n_cases"] = np.array(len(cases))
    # GardnerTED on a C4FM-like real waveform (10 sps), two blocks
    rng = np.random.default_rng(1600)
    sym = rng.choice([-3.0, -1.0, 1.0, 3.0], size=700)
    x = np.repeat(sym, 10).astype(np.float64)
    x = np.convolve(x, np.hanning(15) / np.sum(np.hanning(15)), mode="same") + 0.05 * rng.standard_normal(x.size)
    x = x.astype(np.float32)
    g = GardnerTED(samples_per_symbol=10.0)
    s1, e1 = g.process_block(x[:3000])
    s2, e2 = g.process_block(x[3000:])
    out["g_sha"] = np.array(S.sha256(x))
    out["g_sym"] = np.concatenate([s1, s2])
    out["g_err"] = np.concatenate([e1, e2])
    out["g_counts"] = np.array([len(s1), len(s2)], dtype=np.int64)
    g2 = GardnerTED(samples_per_symbol=10.4166666666666661)
    s3, e3 = g2.process_block(x)
    out["g2_sym"], out["g2_err"] = s3, e3
    save("cqpsk", **out)


def gen_cqpsk_parts():
    """A12 parts as standalone drop-ins: CostasLoop (dsp/p25/cqpsk.py:84-196) and MuellerMullerTED
    (dsp/p25/symbol_timing.py:214-380) on a pi/4-DQPSK carrier, two calls each (carried state)."""
    from wavecapsdr.dsp.p25.cqpsk import CostasLoop
    from wavecapsdr.dsp.p25.symbol_timing import MuellerMullerTED

    fs, sr, n, seed = 48000, 12000, 6000, 1700
    iq, _ = S.dqpsk_iq(n, fs, seed, symbol_rate=sr, snr_db=22.0, freq_offset_hz=60.0)
    x = iq.astype(np.complex128)
    cl = CostasLoop()
    c1, c2 = cl.process_block(x[:3500]), cl.process_block(x[3500:])
    mm = MuellerMullerTED(fs / sr)
    a, b = mm.process_block(np.concatenate([c1, c2])[:2500]), mm.process_block(np.concatenate([c1, c2])[2500:])
    save("cqpsk_parts", args=np.array([fs, sr, n, seed], dtype=np.int64), sha=np.array(S.sha256(iq)),
         costas=np.concatenate([c1, c2]), costas_freq=np.array([cl.frequency_offset]),
         mm_counts=np.array([len(a[0]), len(b[0])], dtype=np.int64),
         mm_sym=np.concatenate([a[0], b[0]]), mm_dec=np.concatenate([a[1], b[1]]), mm_err=np.concatenate([a[2], b[2]]))


def gen_framer():
    """N2: P25P1SoftSyncDetector.process_batch (decoders/p25_framer.py:192-231); N4: pack_f32 (capture.py:134-144)."""
    from wavecapsdr.decoders.p25_framer import P25P1SoftSyncDetector

    rng = np.random.default_rng(1700)
    dib = S.c4fm_frames_dibits(2000, 1701, frame_len=360)
    soft = (np.array([1.0, 3.0, -1.0, -3.0])[dib] + 0.3 * rng.standard_normal(dib.size)).astype(np.float32)
    det = P25P1SoftSyncDetector()
    lens = [700, 5, 1, 0, 23, 24, 25, 1222]
    pos, outs = 0, []
    for n in lens:
        outs.append(det.process_batch(soft[pos:pos + n]))
        pos += n
    assert pos == soft.size
    det1 = P25P1SoftSyncDetector()
    single = np.array([det1.process(float(v)) for v in soft[:64]], dtype=np.float32)
    vals = np.concatenate([np.array([1.0, -1.0, 1.5, -1.5, 0.0, -0.0, np.nan, np.inf, -np.inf, 1.0000001], dtype=np.float32),
                           rng.uniform(-1.3, 1.3, 2038).astype(np.float32)])
    f32 = np.frombuffer(rc.pack_f32(vals.copy()), dtype=np.float32)
    save("framer", soft=soft, lens=np.array(lens, dtype=np.int64), scores=np.concatenate(outs), single=single,
         pattern=det.SYNC_PATTERN_SYMBOLS, f32_in=vals, f32_out=f32)


def gen_lsm():
    """A12 (LSM): decoders/p25.py:190-669 CQPSKDemodulator on streamed, ragged calls.

    The reference's Gardner loop never settles on a pi/4-DQPSK test signal (it emits ~4.7 % more symbols than
    were sent: the clock slips continuously), which makes the symbol stream chaotic: a 1-ulp change (SVML vs
    libm atan2f, the BLAS kernel behind np.convolve) first flips a dibit after ~4 600-12 000 symbols.  Each
    case therefore restarts a fresh demodulator and stays ~2 500 symbols long.  Even so, on a scan of 20 seeds
    about one case in three hit such a flip (the quantisation of mu to 1/128 steps, p25.py:341, is the
    amplifier) in at least one of the oracle's two math flavours; the seeds below are ones where neither
    flavour does -- the reference's own output is only reproducible across hosts in that sense."""
    from wavecapsdr.decoders.p25 import CQPSKDemodulator as RefLSM

    out = {}
    cases = [
        # fs, symbol_rate, n, seed, snr_db, freq_offset_hz, (chunk lo, hi), special
        (48000, 4800, 25000, 1800, 25.0, 40.0, (900, 1700), ""),
        (48000, 4800, 25000, 1801, 12.0, -60.0, (400, 2600), "tiny"),      # includes chunks < 63 samples
        (19200, 4800, 10000, 1806, 25.0, -30.0, (300, 700), ""),
        (48000, 6000, 20000, 1809, 20.0, 20.0, (900, 1700), "silence"),    # Phase-2 rate, zeros in the middle
        (25000, 4800, 13000, 1804, 25.0, 10.0, (500, 900), ""),            # fractional sps 5.208
        (144000, 4800, 60000, 1805, 25.0, 25.0, (2500, 5000), ""),         # round(sps)+4 >= 32: no Gardner, f64 clock
    ]
    for ci, (fs, sr, n, seed, snr, off, (lo, hi), special) in enumerate(cases):
        sps_i = int(round(fs / sr))
        x, _ = S.dqpsk_iq(n, sps_i * sr, seed, symbol_rate=sr, snr_db=snr, freq_offset_hz=off)
        rng = np.random.default_rng(seed + 100)
        lens = []
        while sum(lens) < n:
            lens.append(int(rng.integers(lo, hi)))
            if special == "tiny" and len(lens) % 4 == 2:
                lens.append(int(rng.integers(1, 62)))
        lens[-1] -= sum(lens) - n
        if lens[-1] <= 0:
            lens.pop()
            lens[-1] += n - sum(lens)
        if special == "silence":
            x = x.copy()
            x[8000:9500] = 0
        d = RefLSM(fs, sr)
        pos, dib, st, cnt = 0, [], [], []
        for ln in lens:
            o = d.demodulate(x[pos:pos + ln])
            pos += ln
            dib.append(o)
            cnt.append(len(o))
            st.append([float(d._agc_gain), float(d._freq_offset), float(d._phase_acc), float(d._symbol_clock),
                       float(np.real(d._prev_symbol)), float(np.imag(d._prev_symbol))])
        assert pos == n
        out[f"c{ci}_args"] = np.array([fs, sr, n, seed, snr, off], dtype=np.float64)
        out[f"c{ci}_special"] = np.array(special)
        out[f"c{ci}_sha"] = np.array(S.sha256(x))
        out[f"c{ci}_lens"] = np.array(lens, dtype=np.int64)
        out[f"c{ci}_dibits"] = np.concatenate(dib)
        out[f"c{ci}_counts"] = np.array(cnt, dtype=np.int64)
        out[f"c{ci}_phases"] = np.array(d._symbol_values, dtype=np.float32)
        out[f"c{ci}_state"] = np.array(st, dtype=np.float64)
        out[f"c{ci}_clock_is_f32"] = np.array(isinstance(d._symbol_clock, np.float32))
        if ci == 0:
            out["mmse"] = d._mmse_taps
        out[f"c{ci}_lpf"] = d._baseband_taps
    out["n_cases"] = np.array(len(cases))
    save("lsm", **out)


ALL = dict(zero=gen_zero, caprate=gen_caprate, c4fm_big=gen_c4fm_big, cqpsk_big=gen_cqpsk_big, status=gen_status, cqpsk_parts=gen_cqpsk_parts, chain4=gen_chain4, blanker=gen_blanker, nid=gen_nid, classifier=gen_classifier, recorder=gen_recorder, nr=gen_nr, sam=gen_sam, rawdig=gen_rawdig, lsm=gen_lsm, framer=gen_framer, cqpsk=gen_cqpsk, trunk=gen_trunk, a1=gen_a1, a2=gen_a2, a3=gen_a3, a6=gen_a6, chain=gen_chain, a7=gen_a7, a8=gen_a8, c4fm=gen_c4fm)

if __name__ == "__main__":
    import logging
    logging.disable(logging.CRITICAL)
    names = sys.argv[1:] or list(ALL)
    for nm in names:
        ALL[nm]()
