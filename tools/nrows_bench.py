#!/usr/bin/env python3
"""Throughput of the smaller rows (N2 soft sync / BCH, LSM and Phase-2 CQPSK banks) on synthetic input.  Diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WAVEHIP_PKG_DIR", os.path.join(ROOT, "wavecap-sdr_amd")), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# N2: soft sync correlation, 64 channels x 10 s of symbols per call
C, n = 64, 48000
soft = torch.randn(C, n, device="cuda").mul_(2.0)
bank = wavehip.SoftSyncBank(C)
el = timeit(lambda: bank.process_device(soft))
print(f"soft sync: {C} channels x {n} symbols in {el*1e3:.3f} ms = {C*n/el/1e9:.2f} G symbols/s ({n/4800/el:.0f} x real time per channel)", flush=True)

# N2: BCH(63,16,23) exhaustive decode
from wavehip.fec import BCHDecoder
dec = BCHDecoder()
for nw in (64, 4096):
    w = torch.randint(0, 1 << 62, (nw,), dtype=torch.int64, device="cuda")
    el = timeit(lambda: dec.decode_device(w))
    print(f"BCH(63,16,23): {nw} words in {el*1e3:.3f} ms = {nw/el/1e6:.2f} M words/s", flush=True)

# A12: LSM bank (Phase-1 CQPSK, 19 200 S/s) and Phase-2 CQPSK bank (48 000 S/s), 64 channels x 1 s per call
for name, mk, fs in (("LSM bank", lambda: wavehip.LSMBank(64, 19200, 4800, max_samples_per_call=19200), 19200),
                     ("CQPSK bank", lambda: wavehip.CQPSKBank(64, 48000, 12000, max_samples_per_call=48000), 48000)):
    b = mk()
    x = torch.view_as_complex(torch.randn(64, fs, 2, device="cuda").mul_(0.3))
    el = timeit(lambda: b.demodulate_device(x), reps=5)
    print(f"{name}: 64 channels x 1 s ({fs} S/s) in {el*1e3:.2f} ms = {1.0/el:.0f} x real time per channel, {64*fs/el/1e6:.1f} MS/s x channels", flush=True)
