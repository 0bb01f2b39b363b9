#!/usr/bin/env python3
"""NBFM bank timing (BASELINE configs[1]) -- diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip
fs, n, K, chunks = 2_400_000, 120_000, 32, 200
offs = S.nbfm_bank_offsets(K)
cfgs = [wavehip.ChannelConfig(mode="nbfm", offset_hz=o, enable_deemphasis=False) for o in offs]
bank = wavehip.ChannelBank(fs, n, cfgs, input_format="int16")
d_in = torch.from_numpy(np.tile(S.pack_iq16_np(S.nbfm_bank_c64(n, fs, seed=2)), chunks)).cuda()
audio = torch.empty((chunks, K, bank.n_out), dtype=torch.float32, device="cuda")
met = torch.empty((chunks, K, 4), dtype=torch.float32, device="cuda")
for _ in range(3): bank.process_device(d_in, chunks, audio, met)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): bank.process_device(d_in, chunks, audio, met)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
print(f"nbfm bank: {el*1e3:.3f} ms per launch -> {chunks*n/el/1e6*K:.0f} MS/s x channels")
