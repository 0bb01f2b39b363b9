#!/usr/bin/env python3
"""One setting of the 1024-channel filterbank, N launches (for rocprofv3 counter passes):
V=5 GPW=3 MAP=0 FMT=cf32|int16 N=12 LOG2N=28.  Diagnostics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WAVEHIP_PKG_DIR", os.path.join(ROOT, "wavecap-sdr_amd"))]   # WAVEHIP_PKG_DIR: a scratch (DIAG) build
import torch, wavehip
n = 1 << int(os.environ.get("LOG2N", "28"))
fmt = os.environ.get("FMT", "cf32")
if fmt == "int16": x = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda")
else: x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
ch = wavehip.PolyphaseChannelizer(10_000_000, 9765)
ch.tune(prefetch=int(os.environ.get("V", "5")), hops_per_run=int(os.environ.get("GPW", "3")), run_map=int(os.environ.get("MAP", "0")), alt_dir=int(os.environ.get("ALT", "1")))
ch.profile(True)
out = torch.empty((ch.hops(n), 1024), dtype=torch.complex64, device="cuda")
t = []
for i in range(int(os.environ.get("N", "12"))):
    ch.process_device(x, out); t.append(ch.last_kernel_ms())
t = sorted(t[2:])
print(f"V={os.environ.get('V','5')} GPW={os.environ.get('GPW','3')} MAP={os.environ.get('MAP','0')} {fmt}: median {t[len(t)//2]:.4f} ms min {t[0]:.4f}")
