#!/usr/bin/env python3
"""1024-channel filterbank, int16 input: prefetch forms and run lengths side by side in one process (interleaved rounds).
Diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
n = 1 << 28
x = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda")
variants = [("dma", dict(prefetch=3)), ("regs", dict(prefetch=1)), ("dma, 256-hop runs", dict(prefetch=3, hops_per_run=64)),
            ("shaped kernel", dict(path="shaped"))]
# (measured with two more forms, since removed: `nt` on the global_load_lds -- 1.142-1.154 ms against 1.145-1.158 for the
#  default policy, no difference; 8-byte int16 pair loads into registers with a lane-pair exchange -- 1.273-1.284 ms against
#  1.263-1.270 for the 4-byte register loads and 1.107-1.175 for the DMA form on that box: load width is not what the register
#  form lacks)
chs = [(name, wavehip.PolyphaseChannelizer(10_000_000, 9765).tune(**kw)) for name, kw in variants]
out = torch.empty((chs[0][1].hops(n), 1024), dtype=torch.complex64, device="cuda")
ref = None
for name, ch in chs:
    for _ in range(5): ch.process_device(x, out)
    if ref is None: ref = out.clone()
    else: print(name, "bit-equal to dma:", bool(torch.equal(out, ref)), flush=True)
for rnd in range(3):
    line = []
    for name, ch in chs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ch.process_device(x, out)
        torch.cuda.synchronize()
        line.append(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.4f} ms")
    print("  ".join(line), flush=True)
