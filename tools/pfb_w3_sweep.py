#!/usr/bin/env python3
"""1024-channel filterbank: kernel form x run length (groups of 4 hops per workgroup) x run -> workgroup mapping, kernel time
by HIP events, interleaved rounds in one process; every setting's output is compared with the first one's (bit-equal).
VARIANTS=3,5  GPWS=0,2,3,4,6,8  MAPS=0,-1,4,16  ALTS=0,1  FMT=cf32|int16|both  ROUNDS=9.  Diagnostics."""
import os, sys, statistics, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WAVEHIP_PKG_DIR", os.path.join(ROOT, "wavecap-sdr_amd"))]   # WAVEHIP_PKG_DIR: a scratch (DIAG) build
import torch, wavehip
n = 1 << int(os.environ.get("LOG2N", "28"))
ints = lambda k, d: [int(v) for v in os.environ.get(k, d).split(",")]
variants, gpws, maps = ints("VARIANTS", "3,5"), ints("GPWS", "0,2,3,4,6,8"), ints("MAPS", "0,-1,4,16")
alts = ints("ALTS", "1")
rounds = int(os.environ.get("ROUNDS", "9"))
fmts = os.environ.get("FMT", "both")
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
x16 = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda")
ch = wavehip.PolyphaseChannelizer(10_000_000, 9765); ch.profile(True)
H = ch.hops(n)
out = torch.empty((H, 1024), dtype=torch.complex64, device="cuda")
ref = torch.empty((H, 1024), dtype=torch.complex64, device="cuda")
for name, inp, by in (("cf32", x, 24), ("int16", x16, 20)):
    if fmts not in ("both", name): continue
    settings = list(itertools.product(variants, gpws, maps, alts))
    t = {s: [] for s in settings}
    bad = []
    for rnd in range(rounds):
        for s in settings:
            v, g, m, al = s
            ch.tune(prefetch=v, hops_per_run=g, run_map=m, alt_dir=al); ch.reset()
            first = rnd == 0 and s == settings[0]
            ch.process_device(inp, ref if first else out)
            k = ch.last_kernel_ms()
            if rnd == 0 and not first and not torch.equal(out, ref): bad.append(s)
            if rnd > 0: t[s].append(k)
    print(f"== {name}: {len(settings)} settings, {rounds - 1} timed rounds; outputs differing from the first setting: {bad}")
    for s in sorted(settings, key=lambda s: statistics.median(t[s])):
        med = statistics.median(t[s])
        print(f"{name} v{s[0]} gpw={s[1]:3d} map={s[2]:3d} alt={s[3]}: {med:.4f} ms (min {min(t[s]):.4f})  {by * n / (med * 1e-3) / 8e12:.4f} of 8 TB/s",
              flush=True)
