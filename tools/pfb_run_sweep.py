#!/usr/bin/env python3
"""Run-kernel tuning: hops per wave (tune(hops_per_run=...)) x input size, for channel counts the run kernel takes.  Diagnostics only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip

for fs, bw in ((8_000_000, 25_000), (3_200_000, 12_500), (6_400_000, 12_500)):
    for logn in (23, 24, 26):
        n = 1 << logn
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
        line = []
        for hpw in (0, 8, 12, 16, 24, 32, 48, 64, 96):
            ch = wavehip.PolyphaseChannelizer(fs, bw).tune(path="run", hops_per_run=hpw)
            M = ch.channel_count
            out = torch.empty((ch.hops(n), M), dtype=torch.complex64, device="cuda")
            for _ in range(2): ch.process_device(x, out)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): ch.process_device(x, out)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
            line.append(f"{hpw}:{el*1e6:7.1f}")
        print(f"M={M} n=2^{logn} (floor {n*24/5e12*1e6:6.1f} us @5TB/s)  us per call by hops/wave  " + "  ".join(line), flush=True)
