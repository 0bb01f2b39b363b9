#!/usr/bin/env python3
"""One WBFM row through chan_rows_kernel with 1/2/4/8 waves per row (WH_ROWS_WAVES at run time): per-sample cost and
fixed overhead of the time-parallel rows kernel.  Diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip
fs, n = 2_400_000, 120_000
bank = wavehip.ChannelBank(fs, n, [wavehip.ChannelConfig(mode="wbfm", offset_hz=0.0)])
d_in = torch.from_numpy(S.noise_c64(n, 3)).cuda()
for w in (1, 2, 4, 8):
    os.environ["WH_ROWS_WAVES"] = str(w)
    for _ in range(3): bank.process_device(d_in, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): bank.process_device(d_in, 1)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
    depth = (n + 64 * w - 1) // (64 * w) + 2564
    print(f"waves/row {w}: {el*1e3:7.3f} ms per chunk; depth {depth} samples per lane -> {el*1e9/depth:6.1f} ns per sample if all of it", flush=True)
