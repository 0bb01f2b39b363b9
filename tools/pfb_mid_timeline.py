#!/usr/bin/env python3
"""Shaped filterbank kernel, per-wave phase timeline of one workgroup (needs a library built with `make DIAG=1`):
cycle stamps at loop top / after the arm MAC / after the first barrier / after the passes, per wave and group."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
from wavehip import _lib
fs, bw = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8_000_000, 25_000)
n = 1 << int(os.environ.get("LOGN", "26"))
hpr = int(os.environ.get("HPR", "0"))
ch = wavehip.PolyphaseChannelizer(fs, bw).tune(hops_per_run=hpr)
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
out = torch.empty((ch.hops(n), ch.channel_count), dtype=torch.complex64, device="cuda")
st = torch.zeros((8, 64, 4), dtype=torch.int64, device="cuda")
f = _lib.lib.wh_diag_mid_stamps
f.argtypes = [C.c_void_p, C.c_int]; f.restype = None
for abl in (0, 3):
    _lib.check(_lib.lib.wh_pfb_tune(ch._h, 4, abl), "tune")
    for wg in (5, int(os.environ.get('WG2', '700'))):
        f(None, 0)
        for _ in range(2): ch.process_device(x, out)
        st.zero_(); f(st.data_ptr(), wg)
        ch.process_device(x, out); torch.cuda.synchronize()
        s = st.cpu().numpy()
        t0 = s[s > 0].min()
        print(f"--- ablate={abl} workgroup {wg}: per group, per wave: [top, +mac, +barrier, +passes] in cycles from the first stamp; wave 4 has no passes")
        for g in range(0, 10):
            row = []
            for w in range(5):
                v = s[w, g]
                if v[0] == 0: continue
                row.append(f"w{w}: {v[0]-t0:7d} +{v[1]-v[0]:5d} +{v[2]-v[1]:5d} +{v[3]-v[2]:5d}")
            print(f"g{g:2d}  " + " | ".join(row))
f(None, 0)
