#!/usr/bin/env python3
"""wh_diag_stream_1r2w variants (WH_DIAG_MODE, WH_DIAG_BLOCKS -- read by a library built with `make DIAG=1` only): which plain
streaming kernel is the right yardstick."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
from wavehip import _lib
n = 1 << 28
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
out = torch.empty(2 * n + 4096, dtype=torch.complex64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mode in (0, 1, 2):
    for blocks in (1024, 2048, 4096, 8192, 16384, 65536, 1 << 20):
        os.environ["WH_DIAG_MODE"], os.environ["WH_DIAG_BLOCKS"] = str(mode), str(blocks)
        t = []
        for _ in range(6):
            ev0.record()
            _lib.check(_lib.lib.wh_diag_stream_1r2w(x.data_ptr(), out.data_ptr(), n, _lib.stream_ptr(torch)), "diag")
            ev1.record(); torch.cuda.synchronize(); t.append(ev0.elapsed_time(ev1))
        ms = statistics.median(t[2:])
        print(f"mode {mode} blocks {blocks:8d}: {ms:7.4f} ms {n*24/ms/1e6:7.0f} GB/s", flush=True)
