"""The batched seam (wavehip.process_channels_parallel, drop-in for Capture._process_channels_parallel,
capture.py:2489-2597) with stand-ins for Capture / Channel that carry exactly the attributes the seam touches.  The first
test re-expresses the reference's own seam test (backend/tests/unit/test_capture_dsp_timeout.py:10-29: a slow operator
and a 10 ms timeout must give (channel, None) quickly); oracle/check_boundary.py runs the same scenario against the
real reference Capture in the build container."""

import threading
import time
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace

import numpy as np
import pytest


class FakeChannel:
    def __init__(self, cid, mode="wbfm", state="running"):
        self.cfg = SimpleNamespace(id=cid, capture_id="c1", mode=mode, offset_hz=0.0)
        self.state = state
        self.rssi_db = None
        self.signal_power_db = None
        self.audio_seen = []

    def _update_audio_metrics(self, audio):
        self.audio_seen.append(audio)


class FakeCapture:
    def __init__(self, channels):
        self.cfg = SimpleNamespace(id="c1", sample_rate=1_000_000)
        self._channels = {c.cfg.id: c for c in channels}
        self._dsp_inflight_lock = threading.Lock()
        self._dsp_inflight = 0
        self._dsp_drop_last_log = 0.0
        self.stateful_calls = []

    def _apply_stateful_processing(self, ch, audio, samples):
        self.stateful_calls.append(ch.cfg.id)
        if ch.cfg.id == "boom":
            raise RuntimeError("decoder failed")
        return audio


@pytest.fixture()
def seam():
    import wavehip.capture_seam as s
    return s


def test_process_channels_parallel_timeout(seam, monkeypatch):
    channel = FakeChannel("ch1")
    capture = FakeCapture([channel])

    def slow_dsp(_capture, _samples, _cfgs):
        time.sleep(0.1)
        return [(None, {})]

    monkeypatch.setattr(seam, "dispatch_chunk", slow_dsp)
    samples = np.zeros(1000, dtype=np.complex64)
    with ThreadPoolExecutor(max_workers=1) as executor:
        start = time.perf_counter()
        results = seam.process_channels_parallel(capture, samples, executor, timeout=0.01)
        elapsed = time.perf_counter() - start
    assert len(results) == 1
    result_channel, result_audio = results[0]
    assert result_channel is channel
    assert result_audio is None
    assert elapsed < 0.2
    assert capture._dsp_inflight == 0      # released when the late job ended


def test_results_order_metrics_stateful_tail_and_errors(seam, monkeypatch):
    chans = [FakeChannel("a"), FakeChannel("stopped", state="stopped"), FakeChannel("boom"), FakeChannel("b")]
    capture = FakeCapture(chans)
    audio_a, audio_b = np.ones(8, np.float32), np.full(8, 2.0, np.float32)

    def dsp(_capture, _samples, cfgs):
        assert [c.id for c in cfgs] == ["a", "boom", "b"]          # running channels only, capture order
        return [(audio_a, {"rssi_db": -30.0, "signal_power_db": -12.0}), (np.zeros(8, np.float32), {"rssi_db": -50.0}),
                (audio_b, {"rssi_db": -40.0})]

    monkeypatch.setattr(seam, "dispatch_chunk", dsp)
    with ThreadPoolExecutor(max_workers=3) as ex:
        res = seam.process_channels_parallel(capture, np.zeros(100, np.complex64), ex, timeout=1.0)
    assert [(c.cfg.id, a is not None) for c, a in res] == [("a", True), ("boom", False), ("b", True)]
    assert res[0][1] is audio_a and res[2][1] is audio_b
    assert chans[0].rssi_db == -30.0 and chans[0].signal_power_db == -12.0 and chans[3].rssi_db == -40.0
    assert chans[3].signal_power_db is None
    assert capture.stateful_calls == ["a", "boom", "b"] and len(chans[0].audio_seen) == 1 and not chans[2].audio_seen


def test_backlog_skips_the_cycle_and_operator_exception_drops_audio(seam, monkeypatch):
    capture = FakeCapture([FakeChannel("a")])
    capture._dsp_inflight = 8                         # max(4, 2 x 3 workers) = 6 already in flight
    with ThreadPoolExecutor(max_workers=3) as ex:
        assert seam.process_channels_parallel(capture, np.zeros(10, np.complex64), ex) == []
    capture._dsp_inflight = 0

    def broken(_c, _s, _cfgs):
        raise RuntimeError("launch failed")

    monkeypatch.setattr(seam, "dispatch_chunk", broken)
    with ThreadPoolExecutor(max_workers=3) as ex:
        res = seam.process_channels_parallel(capture, np.zeros(10, np.complex64), ex)
    assert len(res) == 1 and res[0][1] is None and capture._dsp_inflight == 0
    assert seam.process_channels_parallel(FakeCapture([]), np.zeros(10, np.complex64), None) == []


@pytest.mark.gpu
def test_seam_runs_the_batched_dispatcher_on_the_device():
    """Through the real dispatcher: three NBFM channels + one digital-voice channel of one capture, one upload; audio equal
    to the single-channel operator's, metrics on the channel objects."""
    import signals as S
    import wavehip

    fs, n = 2_400_000, 120_000
    offs = S.nbfm_bank_offsets(4)
    iq = S.nbfm_bank_c64(n, fs, seed=21, n_ch=4)
    chans = []
    for k in range(3):
        c = FakeChannel(f"n{k}")
        c.cfg = wavehip.ChannelConfig(id=f"n{k}", mode="nbfm", offset_hz=offs[k], enable_deemphasis=False)
        chans.append(c)
    d = FakeChannel("p25")
    d.cfg = wavehip.ChannelConfig(id="p25", mode="p25", offset_hz=offs[3])
    chans.append(d)
    capture = FakeCapture(chans)
    capture.cfg.sample_rate = fs
    with ThreadPoolExecutor(max_workers=3) as ex:
        res = wavehip.process_channels_parallel(capture, iq, ex, timeout=30.0)
    assert [c.cfg.id for c, _ in res] == ["n0", "n1", "n2", "p25"]
    for k in range(3):
        a_ref, m_ref = wavehip.process_channel_dsp_stateless(iq, fs, chans[k].cfg)
        assert np.array_equal(res[k][1], a_ref) and chans[k].rssi_db == m_ref["rssi_db"]
    assert res[3][1] is None and chans[3].rssi_db is not None


@pytest.mark.gpu
def test_wire_formats_written_by_the_finalize_kernel():
    """ChannelBank(..., wire="pcm16" | "f32"): the int16 / clipped-float32 buffers the finalize kernel writes are
    byte-identical to pack_pcm16 / pack_f32 (capture.py:119-144 rules, pinned by the A1 / N4 goldens) of the float32
    audio of the same call -- squelched rows and a 200-chunk batch included."""
    import torch
    import signals as S
    import wavehip

    fs, n, K = 2_400_000, 120_000, 4
    offs = S.nbfm_bank_offsets(K)
    i16 = S.pack_iq16_np(S.nbfm_bank_c64(n, fs, seed=22, n_ch=K))
    cfgs = [wavehip.ChannelConfig(mode="nbfm", offset_hz=o, enable_deemphasis=False,
                                  squelch_db=(0.0 if k == 2 else None)) for k, o in enumerate(offs)]
    bank = wavehip.ChannelBank(fs, n, cfgs, input_format="int16", apply_squelch=True)
    d = torch.from_numpy(np.tile(i16, 3)).cuda()
    for wire, pack in (("pcm16", wavehip.pack_pcm16), ("f32", wavehip.pack_f32)):
        audio, met, w = bank.process_device(d, 3, wire=wire)
        a = audio.cpu().numpy()
        assert not a[:, 2].any() and a[:, 0].any()                      # channel 2 squelched (rssi < 0 dB), others not
        assert w.cpu().numpy().tobytes() == pack(a.reshape(-1))
        assert torch.equal(audio, bank.process_device(d, 3)[0])          # the float32 audio is unchanged by the wire pass
    res = bank.process_wire(i16, "pcm16")
    ref = bank.process(i16)
    for (wb, m), (a, m2) in zip(res, ref):
        assert wb == wavehip.pack_pcm16(a) and m == m2


@pytest.mark.gpu
def test_overlapping_chunks_share_one_dispatcher_safely(seam, monkeypatch):
    """After a timeout the seam cannot cancel a chunk that is already running, and the next chunk goes to another worker
    of the same executor: two jobs of one capture in flight.  The reference tolerates that because its operator is
    stateless; here the banks own device scratch, so ChannelDispatcher.process serialises per dispatcher.  A first
    chunk held up inside the dispatcher (slow first job, 10 ms timeout -> (channel, None)) and a second chunk submitted
    meanwhile: the second result equals a serial run, the late first job finishes without error."""
    import signals as S
    import wavehip
    from wavehip import channel_ops

    fs, n, K = 2_400_000, 120_000, 4
    offs = S.nbfm_bank_offsets(K)
    iq1 = S.nbfm_bank_c64(n, fs, seed=31, n_ch=K)
    iq2 = S.nbfm_bank_c64(n, fs, seed=32, n_ch=K)
    chans = []
    for k in range(K):
        c = FakeChannel(f"n{k}")
        c.cfg = wavehip.ChannelConfig(id=f"n{k}", mode="nbfm", offset_hz=offs[k], enable_deemphasis=False)
        chans.append(c)
    capture = FakeCapture(chans)
    capture.cfg.sample_rate = fs
    serial = wavehip.ChannelDispatcher(fs).process(iq2, [c.cfg for c in chans])

    gate, entered = threading.Event(), threading.Event()
    real = channel_ops.ChannelBank.process_device
    calls = []

    def slow_first(self, d_in, n_chunks, *a, **kw):
        calls.append(threading.get_ident())
        out = real(self, d_in, n_chunks, *a, **kw)
        if len(calls) == 1:           # the first chunk stalls INSIDE the dispatcher, its launches queued
            entered.set()
            gate.wait(5.0)
        return out

    monkeypatch.setattr(channel_ops.ChannelBank, "process_device", slow_first)
    with ThreadPoolExecutor(max_workers=3) as ex:
        late = seam.process_channels_parallel(capture, iq1, ex, timeout=0.01)
        assert [a for _, a in late] == [None] * K and entered.wait(5.0)
        box = {}
        t = threading.Thread(target=lambda: box.update(res=seam.process_channels_parallel(capture, iq2, ex, timeout=30.0)))
        t.start()
        time.sleep(0.05)
        assert "res" not in box       # the second chunk waits for the dispatcher instead of running beside the first
        gate.set()
        t.join(30.0)
    assert len(calls) == 2 and capture._dsp_inflight == 0
    for (ch, audio), (a_ref, m_ref) in zip(box["res"], serial):
        assert np.array_equal(audio, a_ref) and ch.rssi_db == m_ref["rssi_db"]


@pytest.mark.gpu
def test_retune_reuses_the_bank():
    """offset_hz / squelch_db changes (API PATCH -> capture.py:442-501) must not rebuild a bank: the dispatcher keys banks
    by chain and channel count, a retune is ChannelBank.set_offsets (one 128-byte copy for 32 channels).  32 NBFM
    channels retuned on every chunk: one bank ever created, every chunk's audio and metrics bit-equal to a fresh bank
    built for those offsets, and the retune costs < 0.1 ms per chunk on top of the same loop without it."""
    import torch
    import signals as S
    import wavehip

    fs, n, K = 2_400_000, 120_000, 32
    base = S.nbfm_bank_offsets(K)
    iq = S.nbfm_bank_c64(n, fs, seed=41, n_ch=K)
    disp = wavehip.ChannelDispatcher(fs, apply_squelch=True)
    for step in range(4):
        cfgs = [wavehip.ChannelConfig(mode="nbfm", offset_hz=o + 1000.0 * step * (1 if k % 2 else -1),
                                      enable_deemphasis=False, squelch_db=(-20.0 + step if k == 5 else None))
                for k, o in enumerate(base)]
        got = disp.process(iq, cfgs)
        fresh = wavehip.ChannelBank(fs, n, cfgs, apply_squelch=True).process(iq)
        for (a, m), (fa, fm) in zip(got, fresh):
            assert np.array_equal(a, fa) and m == fm, step
    assert disp.banks_created == 1 and len(disp._banks) == 1
    # cost of a retune per chunk, device-resident loop
    bank = next(iter(disp._banks.values()))
    d_in = torch.from_numpy(iq).cuda()
    audio, met = bank.process_device(d_in, 1)
    offs = [[int(o + 500 * s) for o in base] for s in range(2)]

    def loop(retune: bool, reps: int = 200) -> float:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            if retune:
                bank.set_offsets(offs[i & 1])
            bank.process_device(d_in, 1, audio=audio, metrics=met)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    loop(True, 20)
    plain = min(loop(False) for _ in range(3))
    tuned = min(loop(True) for _ in range(3))
    print(f"retune per chunk: {1e3 * (tuned - plain):.4f} ms on top of {1e3 * plain:.4f} ms")
    assert tuned - plain < 1e-4
