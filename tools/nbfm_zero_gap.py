#!/usr/bin/env python3
"""NBFM chain on int16 input with EXACT zero samples (a muted stretch and isolated zeros): device vs the numpy oracle.
The discriminator of a zero sample hangs on signed zeros of the mixed values in the reference; diagnostics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip
from oracle import ref_np as O
fs, n, K = 2_400_000, 120_000, 4
offs = S.nbfm_bank_offsets(32)[12:16]
i16 = S.pack_iq16_np(S.nbfm_bank_c64(n, fs, seed=77)).copy()
i16[2 * 30000:2 * 30600] = 0                      # 600 muted samples
for k in (5, 5000, 70001, 119999): i16[2 * k:2 * k + 2] = 0
z = i16.astype(np.float32) / 32768.0
z = (z[0::2] + 1j * z[1::2]).astype(np.complex64)
cfgs = [wavehip.ChannelConfig(mode="nbfm", offset_hz=o, enable_deemphasis=False) for o in offs]
res = wavehip.ChannelBank(fs, n, cfgs, input_format="int16").process(i16)
for k, o in enumerate(offs):
    a_ref, m_ref = O.process_channel_nbfm(z, fs, o)
    a, m = res[k]
    if a_ref is None or a is None:
        print(k, "class", a is None, a_ref is None); continue
    print(f"ch {k} off {o}: peak-rel err {np.max(np.abs(a - a_ref)) / np.max(np.abs(a_ref)):.3g}  rssi {m['rssi_db']:.4f} vs {m_ref['rssi_db']:.4f}")
