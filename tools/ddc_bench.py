#!/usr/bin/env python3
"""Trunking front-end bank (row N3): K NCO + two-stage-decimator front-ends on one wideband buffer per call, beside
the oracle's full-rate lfilter path for one front-end.  Diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip
from oracle import ref_np as O

fs = 6_000_000
n = fs // 10                       # 100 ms of wideband IQ per call
x = S.noise_c64(n, 5)
d = torch.from_numpy(x).cuda()
for plan in ("recorder", "control"):
    for K in (1, 8, 32, 64):
        bank = wavehip.TrunkingDDCBank(K, fs, plan=plan, max_samples_per_call=n)
        offs = np.linspace(-2.0e6, 2.0e6, K) if K > 1 else np.array([250e3])
        for _ in range(3): bank.process_device(d, offs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): bank.process_device(d, offs)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
        print(f"{plan:8s} K={K:2d}: D = {bank.stage1_factor} x {bank.stage2_factor} -> {bank.output_rate} S/s; "
              f"{el*1e3:7.3f} ms per 100 ms call = {0.1/el:7.1f} x real time for the bank, "
              f"{K*n/el/1e9:6.2f} G input sample-channels/s", flush=True)
b1 = wavehip.TrunkingDDCBank(1, fs, plan="recorder", max_samples_per_call=n)
ddc = O.TrunkingDDC(fs, b1.stage1_factor, b1.stage2_factor)
t0 = time.perf_counter(); ddc.process(x, 250e3); el = time.perf_counter() - t0
print(f"CPU port, one recorder front-end, one call: {el*1e3:.1f} ms ({0.1/el:.1f} x real time)", flush=True)
