#!/usr/bin/env python3
"""Where the sequential symbol kernels saturate: C4FM bank (one wave per channel in k_seq), Phase-2 CQPSK bank (one wave per
channel in k_cq_seq) and LSM bank (one wave per channel in k_lsm_seq) at 64 / 256 / 1024 / 4096 channels: time per call, x real time
per channel and aggregate sample rate.  Diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import signals as S, wavehip


def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


fs, call = 48000, 4800
base = np.stack([S.c4fm_iq(call * 10, fs, 1000 + k, snr_db=20.0, freq_offset_hz=50.0 * k - 200)[0] for k in range(8)])
for C in (64, 256, 1024, 4096):
    x = torch.from_numpy(np.tile(base, (C // 8, 1))).cuda()
    bank = wavehip.C4FMBank(C, fs, max_samples_per_call=call)
    for s in range(0, call * 10, call):          # lock first (sync events are the expensive part)
        bank.demodulate_device(x[:, s:s + call])
    el = timeit(lambda: bank.demodulate_device(x[:, call * 9: call * 10]))
    print(f"C4FM bank  {C:5d} channels: {el*1e3:8.3f} ms per 100 ms call = {0.1/el:7.1f} x real time per channel, "
          f"{C*call/el/1e6:8.1f} MS/s x channels", flush=True)
    del bank, x
for name, mk, rate in (("CQPSK bank", lambda C: wavehip.CQPSKBank(C, 48000, 12000, max_samples_per_call=48000), 48000),
                       ("LSM bank  ", lambda C: wavehip.LSMBank(C, 19200, 4800, max_samples_per_call=19200), 19200)):
    for C in (64, 256, 1024, 4096):
        b = mk(C)
        z = torch.view_as_complex(torch.randn(C, rate, 2, device="cuda").mul_(0.3))
        el = timeit(lambda: b.demodulate_device(z), reps=3)
        print(f"{name} {C:5d} channels: {el*1e3:8.2f} ms per 1 s call   = {1.0/el:7.1f} x real time per channel, "
              f"{C*rate/el/1e6:8.1f} MS/s x channels", flush=True)
        del b, z
