#!/usr/bin/env python3
"""In-process yardstick: the filterbank's traffic shape (8 B in, 16 B out per sample, 2^28 samples) with no
arithmetic (wh_diag_stream_1r2w) beside the filterbank itself and torch's copy / mul kernels."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
from wavehip import _lib

n = 1 << 28
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
ch = wavehip.PolyphaseChannelizer(10_000_000, 9765)
ch.profile(True)
out = torch.empty((ch.hops(n) + 2, 1024), dtype=torch.complex64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {"stream_1r2w": [], "pfb1024": [], "torch_copy_1r1w": [], "torch_mul_1r1w": []}
y = torch.empty_like(x)
for rnd in range(8):
    ev0.record()
    _lib.check(_lib.lib.wh_diag_stream_1r2w(x.data_ptr(), out.data_ptr(), n, _lib.stream_ptr(torch)), "diag")
    ev1.record(); torch.cuda.synchronize()
    res["stream_1r2w"].append(ev0.elapsed_time(ev1))
    ch.process_device(x, out); res["pfb1024"].append(ch.last_kernel_ms())
    ev0.record(); y.copy_(x); ev1.record(); torch.cuda.synchronize()
    res["torch_copy_1r1w"].append(ev0.elapsed_time(ev1))
    xr, yr = torch.view_as_real(x), torch.view_as_real(y)
    ev0.record(); torch.mul(xr, 2.0, out=yr); ev1.record(); torch.cuda.synchronize()
    res["torch_mul_1r1w"].append(ev0.elapsed_time(ev1))
for k, v in res.items():
    ms = statistics.median(v[2:])
    b = n * (24 if k in ("stream_1r2w", "pfb1024") else 16)
    print(f"{k:18s} {ms:7.4f} ms  {b/ms/1e6:7.0f} GB/s", flush=True)
