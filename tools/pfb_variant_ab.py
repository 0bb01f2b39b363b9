#!/usr/bin/env python3
"""1024-channel filterbank: tune(prefetch=1) (register prefetch) vs 3 (LDS-DMA prefetch, counted waits): same outputs?
kernel time (HIP events) interleaved in one process; cf32 and int16 input.  Diagnostics."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
n = 1 << int(os.environ.get("LOG2N", "28"))
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
x16 = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda")
chs = []
for v in (1, 3):
    ch = wavehip.PolyphaseChannelizer(10_000_000, 9765).tune(prefetch=v); ch.profile(True); chs.append(ch)
out = [torch.empty((chs[0].hops(n), 1024), dtype=torch.complex64, device="cuda") for _ in chs]
for name, inp in (("cf32", x), ("int16", x16)):
    t = [[], []]
    for rnd in range(9):
        for i, ch in enumerate(chs):
            ch.process_device(inp, out[i]); k = ch.last_kernel_ms()
            if rnd > 0: t[i].append(k)
    print(f"{name}: equal outputs {bool(torch.equal(out[0], out[1]))};  register prefetch median {statistics.median(t[0]):.4f} ms "
          f"(min {min(t[0]):.4f}),  LDS-DMA prefetch median {statistics.median(t[1]):.4f} ms (min {min(t[1]):.4f})", flush=True)
# ragged sizes through both, several calls (history), exact equality
for m in (1024 * 9 + 512 * 3, 1024 * 300 + 77, 512 * 4099 + 1024, 1024 * 37):
    a, b = chs[0], chs[1]
    a.reset(); b.reset()
    ok = True
    for part in (x[:m], x[m:2 * m + 5]):
        ok &= bool(torch.equal(a.process_device(part), b.process_device(part)))
    print(f"n={m}: equal {ok}", flush=True)
