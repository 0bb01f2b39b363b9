#!/usr/bin/env python3
"""One WBFM bank launch shape for counter collection (diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip
fs, n, chunks = 2_400_000, 120_000, int(os.environ.get("CHUNKS", "200"))
bank = wavehip.ChannelBank(fs, n, [wavehip.ChannelConfig(mode="wbfm", offset_hz=0.0)])
d_in = torch.from_numpy(np.tile(S.noise_c64(n, 3), chunks)).cuda()
for _ in range(5):
    bank.process_device(d_in, chunks)
torch.cuda.synchronize()
