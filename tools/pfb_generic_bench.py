#!/usr/bin/env python3
"""Generic (any M) filterbank path timing -- diagnostics (the reference's benchmark_dsp.py shape is M = 320)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
for fs, bw in ((8_000_000, 25_000), (8_000_000, 31_250), (2_400_000, 12_500), (10_000_000, 9765)):
    ch = wavehip.PolyphaseChannelizer(fs, bw)
    n = 1 << 24
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
    out = torch.empty((ch.hops(n), ch.channel_count), dtype=torch.complex64, device="cuda")
    for _ in range(2): ch.process_device(x, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ch.process_device(x, out)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5
    print(f"fs={fs} bw={bw} M={ch.channel_count}: {el*1e3:8.3f} ms per 2^24 samples -> {n/el/1e6:9.1f} MS/s ({n/el/fs:7.1f} x real time)", flush=True)
