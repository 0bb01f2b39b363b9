#!/usr/bin/env python3
"""Shaped filterbank kernels (pfb_mid.hip) against the kernel each channel count took before (run kernel for 4 | M <= 512,
per-hop kernel otherwise): time per call and algorithmic TB/s (24 B per input sample).  `--sweep` adds the run-length
sweep for the first shape.  pfb_mid_bench.py [--sweep] [--taps=T] [M ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip

args = [a for a in sys.argv[1:] if not a.startswith("--")]
sweep = "--sweep" in sys.argv
T = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--taps=")), 9)
mk = lambda fs, bw: wavehip.PolyphaseChannelizer(fs, bw, taps_per_channel=T)
MS = [int(a) for a in args] or [320]


def timeit(ch, x, out, reps=10):
    for _ in range(3): ch.process_device(x, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ch.process_device(x, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for M in MS:
    fs, bw = M * 25_000, 25_000
    sh = mk(fs, bw).tune(path='shaped')
    assert sh.channel_count == M
    g = torch.Generator(device="cuda").manual_seed(M)
    for logn in ((24, 26, 28) if M == 1024 else (24, 26)):
        n = 1 << logn
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=g).mul_(0.5))
        out = torch.empty((sh.hops(n), M), dtype=torch.complex64, device="cuda")
        t = timeit(sh, x, out)
        line = f"M={M:5d} T={T} n=2^{logn}: shaped {t*1e6:8.1f} us = {n*24/t/1e12:5.2f} TB/s"
        try:
            if M == 1024 and T == 9:
                old, name = mk(fs, bw), "pfb1024 kernel"
            else:
                old = mk(fs, bw).tune(path="run")
                name = "run kernel"
        except RuntimeError:
            old = mk(fs, bw).tune(path="per_hop")
            name = "per-hop kernel"
        to = timeit(old, x, out, reps=3)
        line += f"   {name} {to*1e6:9.1f} us   x{to/t:.2f}"
        if sweep and M == MS[0]:
            sw = []
            for hpr in (16, 24, 32, 48, 64, 96, 128):
                sh.tune(hops_per_run=hpr)
                sw.append(f"{hpr}:{timeit(sh, x, out)*1e6:7.1f}")
            sh.tune(hops_per_run=0)
            line += "   by hops/run " + " ".join(sw)
        print(line, flush=True)
