#!/usr/bin/env python3
"""Shaped filterbank kernels (pfb_mid.hip) against the run kernel and the per-hop kernel: agreement and time per call,
with a sweep of the run length.  Diagnostics only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip

CASES = [(8_000_000, 25_000)]
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]]


def timeit(ch, x, out, reps=10):
    for _ in range(3): ch.process_device(x, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ch.process_device(x, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for fs, bw in CASES:
    sh = wavehip.PolyphaseChannelizer(fs, bw)
    M = sh.channel_count
    hop = wavehip.PolyphaseChannelizer(fs, bw).tune(path="per_hop")
    g = torch.Generator(device="cuda").manual_seed(M)
    x = torch.view_as_complex(torch.randn(M * 1000 + 17, 2, device="cuda", generator=g).mul_(0.5))
    ya, yb = sh.process_device(x), hop.process_device(x)
    err = ((ya - yb).abs().max() / yb.abs().max()).item()
    print(f"M={M}: shaped vs per-hop kernel peak-relative difference {err:.2e}", flush=True)
    for logn in (24, 26):
        n = 1 << logn
        x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=g).mul_(0.5))
        out = torch.empty((sh.hops(n), M), dtype=torch.complex64, device="cuda")
        line = [f"auto:{timeit(sh, x, out)*1e6:7.1f}"]
        for hpr in (16, 24, 32, 48, 64, 96, 128):
            sh.tune(hops_per_run=hpr)
            line.append(f"{hpr}:{timeit(sh, x, out)*1e6:7.1f}")
        sh.tune(hops_per_run=0)
        try:
            run = wavehip.PolyphaseChannelizer(fs, bw).tune(path="run")
            line.append(f"run-kernel:{timeit(run, x, out)*1e6:7.1f}")
        except RuntimeError:
            pass
        print(f"M={M} n=2^{logn} (floor {n*24/5e12*1e6:6.1f} us @5TB/s) us per call by hops/run  " + "  ".join(line), flush=True)
