"""BASELINE.json full-size configurations, checked through size-independent properties (the oracle
would take minutes to hours at these sizes) plus oracle spot checks on slices."""

import numpy as np
import pytest

import signals as S
from conftest import peak_rel_err

pytestmark = pytest.mark.gpu


def test_config3_pfb_full_size_2p28():
    """1024-channel filterbank on 2^28 samples: every row of a spot-checked window equals the oracle on
    that window's input (hop h depends only on samples (h-8)*512 .. h*512+1023), and a centred tone
    lands in its channel over the whole 524 287-hop output."""
    import torch
    import wavehip
    from oracle import ref_np as O

    n = 1 << 28
    ch = wavehip.PolyphaseChannelizer(10_000_000, 9765)
    g = torch.Generator(device="cuda").manual_seed(33)
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=g).mul_(0.5))
    y = ch.process_device(x)
    assert y.shape == (524287, 1024)
    ref = O.PolyphaseChannelizer(10_000_000, 9765)
    for h0 in (0, 1000, 262144, 524287 - 40):          # first hops (zero history), middle, the ragged tail
        lo = max(0, h0 - 8) * 512
        seg = x[lo: (h0 + 40) * 512 + 1024].cpu().numpy()
        ref.reset()
        r = ref.process(seg)
        off = h0 - lo // 512
        want = r[off: off + 40]
        got = y[h0: h0 + len(want)].cpu().numpy()
        if h0 == 0:
            assert peak_rel_err(got, want) <= 1e-5
        else:   # rows with a full 8-block history inside the segment
            assert peak_rel_err(got[:len(want)], want) <= 1e-5
    del y
    k = 300
    t = torch.arange(n, device="cuda", dtype=torch.int64)
    ph = (2 * np.pi / 1024) * ((t * k) % 1024).to(torch.float64)
    tone = torch.complex(torch.cos(ph), torch.sin(ph)).to(torch.complex64)
    del t, ph
    ch.reset()
    yt = ch.process_device(tone)
    p = (yt[16:].abs() ** 2).mean(0)
    assert int(p.argmax()) == k and p[k].item() > 0.75 * p.sum().item()


def test_config3_int16_dma_prefetch_full_size_bit_equal():
    """The default form of the 1024-channel kernel (three workgroups per CU) prefetches through the LDS DMA and waits with
    a COUNTED vmcnt (the prefetch retires, the sixteen younger output stores stay in flight; MI355X_MICROARCH.md: loads,
    stores and LDS-DMA count together, in issue order) -- int16 input into a double-buffered LDS target, complex64 input
    into a single buffer whose copy is issued after the group's first barrier.  At 2^28 samples the whole chip contends
    for HBM with stores in flight -- the steady state the small tests never reach -- and the whole 524 287 x 1024 output
    plus the carried history must equal the two-workgroup register-prefetch kernel on the unpacked copy bit for bit; a
    second call runs with a no-arithmetic streaming kernel co-resident on another HIP stream to perturb load / store
    timing; then the same for the default form on complex64 input (short and long runs) and the two-workgroup DMA form."""
    import torch
    import wavehip
    from wavehip import _lib

    n = 1 << 28
    g = torch.Generator(device="cuda").manual_seed(44)
    i16 = torch.randint(-12000, 12000, (2 * n,), dtype=torch.int16, device="cuda", generator=g)
    xq = torch.view_as_complex((i16.to(torch.float32) / 32768.0).view(n, 2))    # the A1 unpack rule, exact in float32
    dma = wavehip.PolyphaseChannelizer(10_000_000, 9765)                          # int16 input: the default form
    reg = wavehip.PolyphaseChannelizer(10_000_000, 9765).tune(prefetch=1)         # complex64 input, register prefetch
    ya = dma.process_device(i16)
    yb = reg.process_device(xq)
    assert ya.shape == (524287, 1024) and torch.equal(ya, yb)
    assert np.array_equal(dma.arm_history, reg.arm_history)
    # second call (carried history) with a co-resident streaming kernel on another stream
    side = torch.cuda.Stream()
    m = 1 << 26
    src = torch.view_as_complex(torch.randn(m, 2, device="cuda"))
    dst = torch.empty(2 * m, dtype=torch.complex64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(12):
            _lib.check(_lib.lib.wh_diag_stream_1r2w(src.data_ptr(), dst.data_ptr(), m, _lib.stream_ptr(torch)), "diag")
    ya = dma.process_device(i16, out=ya)
    torch.cuda.synchronize()
    yb = reg.process_device(xq, out=yb)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    assert np.array_equal(dma.arm_history, reg.arm_history)
    # the DMA forms on complex64 input as well: the default (single 16 KB buffer) at its own run length, at 12-hop and
    # 256-hop runs, and the two-workgroup form (double buffer)
    reg.reset()
    yb = reg.process_device(xq, out=yb)
    for kw in (dict(), dict(hops_per_run=3, run_map=-1), dict(hops_per_run=64), dict(hops_per_run=5, alt_dir=0), dict(prefetch=3),
               dict(prefetch=7, hops_per_run=4)):
        ya = wavehip.PolyphaseChannelizer(10_000_000, 9765).tune(**kw).process_device(xq, out=ya)
        assert torch.equal(ya, yb), kw


def test_config2_nbfm_10s_int16_bank():
    """32 NBFM channels, 10 s of a 2.4 MS/s int16 stream in ONE launch (200 chunks): chunks are
    independent (stateless operator), so chunk c of the batch == that chunk run alone; spot-check the
    oracle on two (chunk, channel) pairs."""
    import torch
    import wavehip
    from oracle import ref_np as O

    fs, n, K, chunks = 2_400_000, 120_000, 32, 200
    offs = S.nbfm_bank_offsets(K)
    cfgs = [wavehip.ChannelConfig(mode="nbfm", offset_hz=o, enable_deemphasis=False) for o in offs]
    bank = wavehip.ChannelBank(fs, n, cfgs, input_format="int16")
    parts = [S.pack_iq16_np(S.nbfm_bank_c64(n, fs, seed=900 + c % 4, start=(c % 4) * n)) for c in range(4)]
    i16 = np.concatenate([parts[c % 4] for c in range(chunks)])
    a, m = bank.process_device(torch.from_numpy(i16).cuda(), chunks)
    assert a.shape == (chunks, K, 2400) and bool(torch.isfinite(a).all()) and float(a.abs().max()) <= 0.95 + 1e-6
    for c in (0, 77, 199):
        a1, m1 = bank.process_device(torch.from_numpy(parts[c % 4]).cuda(), 1)
        assert torch.equal(a1[0], a[c]) and torch.allclose(m1[0], m[c], atol=1e-5)
    z = O.unpack_iq16(parts[199 % 4])
    for k in (3, 28):
        ref, met = O.process_channel_nbfm(z, fs, offs[k])
        assert peak_rel_err(a[199, k].cpu().numpy(), ref) <= 1e-5
        assert abs(float(m[199, k, 0]) - met["rssi_db"]) <= 2e-4


def test_config4_c4fm_64ch_10s_bit_exact():
    """BASELINE configs[3] as SURVEY 8(d) specifies it: 64 INDEPENDENT P25 C4FM streams at 48 kHz (seeds 1000+k, offsets
    U(-400, 400) Hz, SNR 20 dB), 10 s each, fed in 100 ms calls -- dibits AND soft symbols of ALL 64 channels
    bit-identical to the C oracle (portable atan2 flavour, pinned to the reference goldens by tests/test_c4fm_oracle.py),
    call by call; every channel locks and produces the nominal symbol count."""
    import torch
    import wavehip
    from oracle.c4fm_c import C4FMDemodulatorRef

    fs, C, call, n = 48000, 64, 4800, 480000
    host, _ = S.config4_streams(C, fs, n)
    xs = torch.from_numpy(host).cuda()
    bank = wavehip.C4FMBank(C, fs, max_samples_per_call=call)
    got_d = [[] for _ in range(C)]
    got_s = [[] for _ in range(C)]
    total = np.zeros(C, dtype=np.int64)
    for s in range(0, n, call):
        d, sf, cnt = bank.demodulate_device(xs[:, s:s + call])
        dc, sc, cc = d.cpu().numpy(), sf.cpu().numpy(), cnt.cpu().numpy()
        total += cc
        for c in range(C):
            got_d[c].append(dc[c, :cc[c]].copy())
            got_s[c].append(sc[c, :cc[c]].copy())
    checked = 0
    for c in range(C):
        ref = C4FMDemodulatorRef(sample_rate=fs, atan_mode=1)
        rd, rs = zip(*(ref.demodulate(host[c, s:s + call]) for s in range(0, n, call)))
        for k in range(len(rd)):      # call by call: counts and contents
            assert np.array_equal(got_d[c][k], rd[k]) and np.array_equal(got_s[c][k], rs[k]), (c, k)
        assert ref.state()["sync_count"] > 100, c
        checked += 1
    assert checked == 64
    assert int(total.min()) >= 47990 and int(total.max()) <= 48010


def test_config1_wbfm_10s_single_channel():
    """BASELINE configs[0]: one default WBFM channel, 10 s of a 2.4 MS/s complex64 stream (200 chunks) in one launch:
    every chunk equals that chunk run alone (stateless operator; the time-parallel IIR form is per row), and the
    numpy oracle agrees on two chunks."""
    import torch
    import wavehip
    from oracle import ref_np as O

    fs, n, chunks = 2_400_000, 120_000, 200
    cfg = wavehip.ChannelConfig(mode="wbfm", offset_hz=0.0)
    bank = wavehip.ChannelBank(fs, n, [cfg])
    parts = [S.fm_tone_c64(n, fs, seed=950 + c, audio_hz=400.0 + 300 * c, noise_amp=0.05) for c in range(4)]
    x = np.concatenate([parts[c % 4] for c in range(chunks)])
    a, m = bank.process_device(torch.from_numpy(x).cuda(), chunks)
    assert a.shape == (chunks, 1, 2400) and bool(torch.isfinite(a).all())
    for c in (0, 101, 199):
        a1, m1 = bank.process_device(torch.from_numpy(parts[c % 4]).cuda(), 1)
        assert torch.equal(a1[0], a[c]) and torch.allclose(m1[0], m[c], atol=1e-5)
    for c in (2, 199):
        ref, met = O.process_channel_wbfm(parts[c % 4], fs, 0.0)
        assert peak_rel_err(a[c, 0].cpu().numpy(), ref) <= 1e-5
        assert abs(float(m[c, 0, 0]) - met["rssi_db"]) <= 2e-4


def test_plan_keeps_bits_and_history():
    """PolyphaseChannelizer.plan(): every run length gives the same output bits and the same carried history; planning
    restores the history it found and reports the candidates it timed."""
    import torch
    import wavehip

    n = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=g).mul_(0.5))
    a, b = wavehip.PolyphaseChannelizer(10_000_000, 9765), wavehip.PolyphaseChannelizer(10_000_000, 9765)
    a.process_device(x[: 1 << 16])
    b.process_device(x[: 1 << 16])
    hist = a.arm_history
    best = a.plan(x, candidates=(0, 8, 32))
    assert best in (0, 8, 32) and set(a.planned["median_ms"]) == {0, 8, 32}
    assert np.array_equal(a.arm_history, hist)
    ya, yb = a.process_device(x), b.process_device(x)
    assert torch.equal(ya, yb) and np.array_equal(a.arm_history, b.arm_history)
