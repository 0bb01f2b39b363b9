"""Parity margin of the fused NBFM bank vs the reference goldens (peak-relative error per case)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import signals as S, wavehip as wh
g = np.load(os.path.join(ROOT, "tests", "golden", "chain_analog.npz"))
fs, n, seed0 = (int(v) for v in g["nbfm_args"])
offs = S.nbfm_bank_offsets()
cfgs = [wh.ChannelConfig(mode="nbfm", offset_hz=o, enable_deemphasis=False, enable_mpx_filter=False) for o in offs]
for fmt in ("int16", "cf32"):
    bank = wh.ChannelBank(fs, n, cfgs, input_format=fmt)
    for chunk in range(2):
        i16 = S.pack_iq16_np(S.nbfm_bank_c64(n, fs, seed=seed0 + chunk, start=chunk * n))
        x = i16 if fmt == "int16" else wh.unpack_iq16(i16)
        res = bank.process(x)
        for k in g["nbfm_k"]:
            ref = g[f"nbfm{chunk}_k{int(k)}_audio"]
            e = np.abs(res[int(k)][0] - ref).max() / np.abs(ref).max()
            print(fmt, chunk, int(k), "peak_rel_err %.2e" % e)
