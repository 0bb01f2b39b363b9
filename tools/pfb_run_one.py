#!/usr/bin/env python3
"""One filterbank shape, a few calls (for rocprofv3 --pmc / --kernel-trace): pfb_run_one.py FS BW LOG2N [ITERS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
fs, bw, logn = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
n = 1 << logn
ch = wavehip.PolyphaseChannelizer(fs, bw)
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
out = torch.empty((ch.hops(n), ch.channel_count), dtype=torch.complex64, device="cuda")
ch.process_device(x, out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): ch.process_device(x, out)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / iters
print(f"M={ch.channel_count} n=2^{logn}: {el*1e6:.1f} us per call, {n*24/el/1e9:.1f} GB/s algorithmic", flush=True)
