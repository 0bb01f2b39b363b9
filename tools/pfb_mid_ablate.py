#!/usr/bin/env python3
"""Shaped filterbank kernel, ablations (needs a library built with `make DIAG=1`): time per call with the output stores,
the prefetch loads or the passes suppressed.  pfb_mid_ablate.py [FS BW [LOG2N]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WAVEHIP_PKG_DIR", os.path.join(ROOT, "wavecap-sdr_amd"))]   # WAVEHIP_PKG_DIR: a DIAG build elsewhere
import torch, wavehip
from wavehip import _lib
fs, bw = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8_000_000, 25_000)
logn = int(sys.argv[3]) if len(sys.argv) > 3 else 26
n = 1 << logn
ch = wavehip.PolyphaseChannelizer(fs, bw)
if ch.channel_count == 1024:
    ch.tune(path="shaped")      # the 1024-channel default is the tuned kernel; the ablation bits live in the shaped one
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
out = torch.empty((ch.hops(n), ch.channel_count), dtype=torch.complex64, device="cuda")
for rep in range(2):
    for bits, name in ((0, "full"), (1, "no stores"), (2, "no prefetch loads"), (3, "no stores, no loads"), (4, "no passes (MAC + loads only)"),
                       (6, "MAC only")):
        _lib.check(_lib.lib.wh_pfb_tune(ch._h, 4, bits), "tune")
        for _ in range(3): ch.process_device(x, out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ch.process_device(x, out)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
        print(f"M={ch.channel_count} n=2^{logn} {name:32s} {el*1e6:8.1f} us", flush=True)
