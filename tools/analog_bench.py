#!/usr/bin/env python3
"""Timing of the unfused analog chains (WBFM / AM / SSB / SAM / filtered NBFM) -- diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip


def run(tag, fs, n, cfgs, chunks, fmt="cf32"):
    bank = wavehip.ChannelBank(fs, n, cfgs, input_format=fmt)
    x = S.noise_c64(n, 3)
    d_in = torch.from_numpy(np.tile(x if fmt == "cf32" else S.pack_iq16_np(x), chunks)).cuda()
    K = len(cfgs)
    audio = torch.empty((chunks, K, bank.n_out), dtype=torch.float32, device="cuda")
    met = torch.empty((chunks, K, 4), dtype=torch.float32, device="cuda")
    for _ in range(2):
        bank.process_device(d_in, chunks, audio, met)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        bank.process_device(d_in, chunks, audio, met)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print(f"{tag:28s} K={K:3d} chunks={chunks:4d}: {el*1e3:8.3f} ms/launch  {chunks*n*K/el/1e6:10.0f} MS/s x ch  "
          f"({chunks*n/fs/el:7.1f} x real time per channel)", flush=True)


fs, n = 2_400_000, 120_000
C = wavehip.ChannelConfig
run("wbfm default", fs, n, [C(mode="wbfm", offset_hz=0.0)], 1)
run("wbfm default", fs, n, [C(mode="wbfm", offset_hz=0.0)], 200)
run("wbfm default", fs, n, [C(mode="wbfm", offset_hz=k * 200e3 - 800e3) for k in range(8)], 25)
run("am agc (250 kS/s)", 250_000, 12_500, [C(mode="am", offset_hz=k * 1e4, enable_agc=True) for k in range(16)], 100)
run("ssb agc (250 kS/s)", 250_000, 12_500, [C(mode="ssb", offset_hz=k * 1e4, enable_agc=True) for k in range(16)], 100)
run("sam (250 kS/s)", 250_000, 12_500, [C(mode="sam", offset_hz=k * 1e4, enable_agc=True) for k in range(16)], 100)
run("am agc only (2.4 MS/s)", fs, n, [C(mode="am", offset_hz=k * 1e5, enable_agc=True, enable_am_highpass=False, enable_am_lowpass=False) for k in range(8)], 25)
run("am agc only (2.4 MS/s)", fs, n, [C(mode="am", offset_hz=1e5, enable_agc=True, enable_am_highpass=False, enable_am_lowpass=False)], 1)
run("nbfm + noise reduction", fs, n, [C(mode="nbfm", offset_hz=k * 25e3, enable_deemphasis=False, enable_noise_reduction=True) for k in range(32)], 20)
