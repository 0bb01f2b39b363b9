"""Print where the spectral-noise-reduction parity error sits (row edges vs interior)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "wavecap-sdr_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import wavehip as wh
from test_oracle_golden import nr_cases

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "chain_nr.npz"))
for tag, mode, fs, iq, off, db in nr_cases(g):
    cfg = wh.ChannelConfig(mode=mode, offset_hz=off)
    if mode == "nbfm":
        cfg.enable_deemphasis = cfg.enable_mpx_filter = cfg.enable_fm_highpass = cfg.enable_fm_lowpass = False
    cfg.enable_noise_reduction = True
    cfg.noise_reduction_db = db
    audio, met = wh.process_channel_dsp_stateless(iq, fs, cfg)
    ref = g[f"{tag}_audio"]
    e = np.abs(audio - ref) / np.max(np.abs(ref))
    idx = np.argsort(e)[::-1][:6]
    print(tag, audio.shape, "max %.2e" % e.max(), "worst idx", idx.tolist(), "interior[8:-8] %.2e" % e[8:-8].max(),
          "interior[64:-64] %.2e" % (e[64:-64].max() if e.size > 200 else -1))
