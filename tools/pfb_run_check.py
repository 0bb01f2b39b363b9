#!/usr/bin/env python3
"""Run-kernel (4 | M, 64 <= M <= 512) vs the one-workgroup-per-hop generic kernel: same numbers? how much faster?
Diagnostics only (tune(path=...) selects the kernel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip

CASES = ((8_000_000, 25_000), (2_400_000, 12_500), (6_000_000, 12_500), (2_400_000, 25_000), (1_600_000, 25_000),
         (3_200_000, 12_500), (6_400_000, 12_500), (12_800_000, 25_000), (3_600_000, 12_500))


def make(fs, bw, run):
    return wavehip.PolyphaseChannelizer(fs, bw).tune(path="run" if run else "per_hop")


for fs, bw in CASES:
    a, b = make(fs, bw, True), make(fs, bw, False)
    M = a.channel_count
    g = torch.Generator(device="cuda").manual_seed(M)
    worst, equal = 0.0, True
    for n in (M * 300 + 17, M * 41 + M // 2 + 3, M * 1000):
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=g).mul_(0.5))
        ya, yb = a.process_device(x).clone(), b.process_device(x).clone()
        assert ya.shape == yb.shape
        equal &= bool(torch.equal(ya, yb))
        worst = max(worst, ((ya - yb).abs().max() / yb.abs().max()).item())
    n = 1 << 24
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=g).mul_(0.5))
    out = torch.empty((a.hops(n), M), dtype=torch.complex64, device="cuda")
    t = []
    for ch in (a, b):
        for _ in range(2): ch.process_device(x, out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ch.process_device(x, out)
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / 5)
    print(f"M={M:4d}: bit-equal={equal} worst rel diff={worst:.2e}  run {t[0]*1e3:7.3f} ms ({n*24/t[0]/1e9:7.1f} GB/s)  "
          f"generic {t[1]*1e3:7.3f} ms  x{t[1]/t[0]:.2f}", flush=True)
