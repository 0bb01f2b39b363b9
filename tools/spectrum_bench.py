#!/usr/bin/env python3
"""Spectrum frames (A8): shaped kernel against the Stockham kernel and the rocFFT engine, 2^26 samples per call in frames
of N: time per call, GS/s and algorithmic TB/s (12 B per sample: 8 in, 4 out).  spectrum_bench.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip

NS = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]
total = 1 << 26
x = torch.view_as_complex(torch.randn(total, 2, device="cuda") * 0.3)
for N in NS:
    frames = total // N
    line = f"N={N:5d} frames={frames:6d}:"
    for eng in ("shaped", "stockham", "rocfft"):
        be = wavehip.HipFFTBackend(N, engine=eng)
        for _ in range(3): be.execute_device(x, frames)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): be.execute_device(x, frames)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        line += f"  {eng} {t*1e6:8.1f} us {total/t/1e9:6.1f} GS/s {total*12/t/1e12:5.2f} TB/s"
    print(line, flush=True)
