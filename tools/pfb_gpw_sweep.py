#!/usr/bin/env python3
"""Run-length sweep of the 1024-channel filterbank kernel, complex64 and int16 input, interleaved rounds in one process
(kernel time by HIP events).  usage: pfb_gpw_sweep.py [gpw ...]   (gpw = groups of 4 hops per workgroup; 0 = default)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, wavehip

n = 1 << int(os.environ.get("LOG2N", "28"))
gpws = [int(a) for a in sys.argv[1:]] or [0, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48, 64]
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
i16 = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda")
chs = {g: wavehip.PolyphaseChannelizer(10_000_000, 9765).tune(hops_per_run=g) for g in gpws}
for c in chs.values():
    c.profile(True)
out = torch.empty((next(iter(chs.values())).hops(n), 1024), dtype=torch.complex64, device="cuda")
res = {(g, f): [] for g in gpws for f in ("cf32", "i16")}
for rnd in range(5):
    for g, ch in chs.items():
        for f, src in (("cf32", x), ("i16", i16)):
            ch.process_device(src, out)
            if rnd:
                res[(g, f)].append(ch.last_kernel_ms())
for g in gpws:
    a, b = statistics.median(res[(g, "cf32")]), statistics.median(res[(g, "i16")])
    print(f"gpw={g:3d} ({4*g if g else 256:4d} hops/run): cf32 {a:.4f} ms = {24*n/a/1e9:.0f} GB/s ({24*n/a/8e9:.3f})   "
          f"int16 {b:.4f} ms = {20*n/b/1e9:.0f} GB/s ({20*n/b/8e9:.3f})", flush=True)
