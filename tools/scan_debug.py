import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import signals as S, wavehip as wh
C = wh.ChannelConfig
iq = S.am_tone_c64(4800, 48000, seed=520, carrier_hz=4800.0, depth=0.8)
cases = {
    "hp+lp noagc": C(mode="am", offset_hz=4800.0, enable_agc=False),
    "hp only": C(mode="am", offset_hz=4800.0, enable_agc=False, enable_am_lowpass=False),
    "lp only": C(mode="am", offset_hz=4800.0, enable_agc=False, enable_am_highpass=False),
    "agc only": C(mode="am", offset_hz=4800.0, enable_agc=True, enable_am_highpass=False, enable_am_lowpass=False),
    "all": C(mode="am", offset_hz=4800.0, enable_agc=True),
}
for tag, cfg in cases.items():
    os.environ.pop("WH_IIR_SEQ", None)
    a = wh.ChannelBank(48000, 4800, [cfg]).process(iq)[0][0]
    os.environ["WH_IIR_SEQ"] = "1"
    b = wh.ChannelBank(48000, 4800, [cfg]).process(iq)[0][0]
    e = np.abs(a - b) / np.abs(b).max()
    bad = np.nonzero(e > 1e-6)[0]
    print(tag, "max rel", e.max(), "first bad", bad[:3], "n bad", bad.size)
