#!/usr/bin/env python3
"""1024-channel filterbank: the two-workgroups-per-CU forms (prefetch 1 = registers, 3 = LDS-DMA) against the
three-workgroups-per-CU forms (5 = LDS-DMA, 7 = registers) -- same outputs? kernel time by HIP events, interleaved rounds in
one process, complex64 and int16 input, at several run lengths (groups of 4 hops per workgroup; 0 = built-in).  Diagnostics."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WAVEHIP_PKG_DIR", os.path.join(ROOT, "wavecap-sdr_amd"))]   # WAVEHIP_PKG_DIR: a scratch (DIAG) build
import torch, wavehip
n = 1 << int(os.environ.get("LOG2N", "28"))
variants = [int(v) for v in os.environ.get("VARIANTS", "1,3,5,7").split(",")]
gpws = [int(v) for v in os.environ.get("GPWS", "0,8,16,32").split(",")]
rounds = int(os.environ.get("ROUNDS", "7"))
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
x16 = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda")
chs = {}
for v in variants:
    ch = wavehip.PolyphaseChannelizer(10_000_000, 9765).tune(prefetch=v); ch.profile(True); chs[v] = ch
H = chs[variants[0]].hops(n)
out = torch.empty((H, 1024), dtype=torch.complex64, device="cuda")
ref = torch.empty((H, 1024), dtype=torch.complex64, device="cuda")
for name, inp in (("cf32", x), ("int16", x16)):
    for gpw in gpws:
        t = {v: [] for v in variants}
        eq = {}
        for rnd in range(rounds):
            for v in variants:
                ch = chs[v].tune(hops_per_run=gpw)
                ch.reset()
                ch.process_device(inp, ref if (rnd == 0 and v == variants[0]) else out)
                k = ch.last_kernel_ms()
                if rnd == 0 and v != variants[0]: eq[v] = bool(torch.equal(out, ref))
                if rnd > 0: t[v].append(k)
        line = "  ".join(f"v{v}: {statistics.median(t[v]):.4f} (min {min(t[v]):.4f}){'' if eq.get(v, True) else ' NOT EQUAL'}"
                         for v in variants)
        by = 24 if name == "cf32" else 20
        best = min(variants, key=lambda v: statistics.median(t[v]))
        print(f"{name} gpw={gpw:3d}: {line}   best v{best} = {by * n / (statistics.median(t[best]) * 1e-3) / 8e12:.4f} of 8 TB/s", flush=True)
# ragged sizes through every form, several calls (history), exact equality against the first
for m in (1024 * 9 + 512 * 3, 1024 * 300 + 77, 512 * 4099 + 1024, 1024 * 37):
    res = []
    for v in variants:
        ch = chs[v].tune(hops_per_run=0); ch.reset()
        res.append([ch.process_device(part).clone() for part in (x[:m], x[m:2 * m + 5])])
    ok = all(torch.equal(a, b) for r in res[1:] for a, b in zip(res[0], r))
    print(f"n={m}: all forms equal {ok}", flush=True)
