#!/usr/bin/env python3
"""Kernel-variant sweep for the fused filterbank (diagnostics; interleaved rounds in ONE process).
usage: python tools/pfb_sweep.py "variant,gpw,ablate" ...   e.g.  1,16,0 1,32,0 1,32,1
(variant = tune(prefetch=...): 0 automatic, 1 registers, 3 LDS DMA, 5 / 7 the three-workgroups forms; gpw = run length in groups of 4 hops,
tune(hops_per_run=gpw); ablate needs a library built with `make DIAG=1` and its WH_PFB_ABLATE bits: 1 = no stores, 2 = DMA copies from cache-resident blocks -- the release
library has no such switch)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WAVEHIP_PKG_DIR", os.path.join(ROOT, "wavecap-sdr_amd"))]   # WAVEHIP_PKG_DIR: a scratch (DIAG) build
import torch
import wavehip

n = 1 << int(os.environ.get("LOG2N", "28"))
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 16, 0), (2, 16, 0)]
chs = []
for var, gpw, abl in cfgs:
    os.environ["WH_PFB_ABLATE"] = str(abl)       # read by a DIAG build only
    ch = wavehip.PolyphaseChannelizer(10_000_000, 9765).tune(prefetch=var, hops_per_run=gpw)
    ch.profile(True)
    chs.append(ch)
out = torch.empty((chs[0].hops(n), 1024), dtype=torch.complex64, device="cuda")
# in-run yardstick: plain device copy of 4 GiB (read 4 + write 4), same process, same box
src = torch.empty(1 << 30, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cp = []
for _ in range(6):
    ev0.record(); dst.copy_(src); ev1.record(); torch.cuda.synchronize()
    cp.append(ev0.elapsed_time(ev1))
print(f"copy yardstick: 8 GiB moved in {statistics.median(cp[1:]):.4f} ms -> {8*(1<<30)/statistics.median(cp[1:])/1e6:.0f} GB/s", flush=True)
del src, dst
times = [[] for _ in chs]
for rnd in range(6):
    for i, ch in enumerate(chs):
        ch.process_device(x, out)
        t = ch.last_kernel_ms()
        if rnd > 0:
            times[i].append(t)
for c, t in zip(cfgs, times):
    med = statistics.median(t)
    print(f"variant={c[0]} gpw={c[1]} ablate={c[2]}: median {med:.4f} ms  min {min(t):.4f}  -> {24*n/med/1e6:.0f} GB/s", flush=True)
