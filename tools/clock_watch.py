#!/usr/bin/env python3
"""Clocks / package power while a kernel runs back to back: rocm-smi is sampled from a second thread of this process while the main thread
keeps the launch queue full.  clock_watch.py [pfb|pfb_int16|stream|nbfm|spectrum]  -- is a kernel power-limited?  Diagnostics."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, wavehip
from wavehip import _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "pfb"


def smi():
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
    pw = re.search(r"Package Power \(W\): ([\d.]+)", o)
    return (int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0)


if mode in ("pfb", "pfb_int16"):
    n = 1 << 28
    x = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda") if mode == "pfb_int16" else \
        torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
    ch = wavehip.PolyphaseChannelizer(10_000_000, 9765)
    out = torch.empty((ch.hops(n), 1024), dtype=torch.complex64, device="cuda")
    launch, per, unit = (lambda: ch.process_device(x, out)), 1.25e-3, "2^28 samples"
elif mode == "stream":
    m = 1 << 28
    src = torch.view_as_complex(torch.randn(m, 2, device="cuda")); dst = torch.empty(2 * m, dtype=torch.complex64, device="cuda")
    launch = lambda: _lib.check(_lib.lib.wh_diag_stream_1r2w(src.data_ptr(), dst.data_ptr(), m, _lib.stream_ptr(torch)), "diag")
    per, unit = 1.2e-3, "2^28 samples (no arithmetic)"
elif mode == "nbfm":
    import signals as S
    fs, n, K, chunks = 2_400_000, 120_000, 32, 200
    cfgs = [wavehip.ChannelConfig(mode="nbfm", offset_hz=o, enable_deemphasis=False) for o in S.nbfm_bank_offsets(K)]
    bank = wavehip.ChannelBank(fs, n, cfgs, input_format="int16")
    d_in = torch.from_numpy(np.tile(S.pack_iq16_np(S.nbfm_bank_c64(n, fs, seed=2)), chunks)).cuda()
    audio = torch.empty((chunks, K, bank.n_out), dtype=torch.float32, device="cuda"); met = torch.empty((chunks, K, 4), dtype=torch.float32, device="cuda")
    launch, per, unit = (lambda: bank.process_device(d_in, chunks, audio, met)), 1.4e-3, "32 ch x 200 chunks"
else:
    N, total = 2048, 1 << 26
    xs = torch.view_as_complex(torch.randn(total, 2, device="cuda") * 0.3)
    be = wavehip.HipFFTBackend(N)
    launch, per, unit = (lambda: be.execute_device(xs, total // N)), 0.17e-3, "2^26 samples in 2048-point frames"
for _ in range(20): launch()
torch.cuda.synchronize()
print(f"{mode}: idle sclk / power {smi()}")
import threading
reps = int(5.0 / per)
samples, stop = [], threading.Event()


def sampler():
    time.sleep(1.5)
    while not stop.is_set():
        samples.append(smi())
        time.sleep(0.4)


th = threading.Thread(target=sampler); th.start()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps): launch()          # (the launch queue's back-pressure keeps this loop in step with the GPU)
ev1.record()
torch.cuda.synchronize()
stop.set(); th.join()
print(f"{mode}: under load (sclk MHz, W): {samples[:8]};  {ev0.elapsed_time(ev1) / reps:.4f} ms per launch of {unit}")
