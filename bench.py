#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X demod hot path.

Workload (BASELINE.json configs[2], the config the metric's roofline is quoted on; configs[4]
when --gpus N > 1): 1024-channel 2x-oversampled polyphase filterbank, fs = 10 MS/s,
channel_bandwidth = 9765 Hz (M = 1024, 9 taps/arm), complex64 input resident in HBM.
One step = one pass of the filterbank over one 2^28-sample buffer of synthetic IQ (seeded
on-device normal noise), followed by the scanner/activity statistics of the last scan window
(1024 hops) and, for N > 1, their RCCL all-reduce.  One independent device stream per GPU
(weak scaling, no data-path collective).

metric value = input MS/s x channels demodulated, whole job.
roofline     = fused pfb1024 kernel: 24 algorithmic bytes per input sample (8 B read + 2 x 8 B
               written) / HIP-event duration of that kernel, vs 8 TB/s HBM peak.
cpu_baseline = the numpy oracle's PolyphaseChannelizer.process on a bounded sample, 1 thread.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "wavecap-sdr_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FS = 10_000_000
BW = 9765
M = 1024
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SAMPLE = 24.0


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_CPU_X = None


def _cpu_pfb_worker(args):
    """One hop range of the oracle's channelizer on a shared input: the block holding the 8 hops of history before the
    range is processed too (its rows are dropped), so every worker returns exactly its own rows."""
    lo, hi, seed, n = args
    import signals as S
    from oracle import ref_np as O

    x = _CPU_X if _CPU_X is not None and len(_CPU_X) == n else S.noise_c64(n, seed)   # shared with forked / threaded workers
    ch = O.PolyphaseChannelizer(FS, BW)
    h0 = max(0, lo - 8)
    seg = x[h0 * 512: (hi - 1) * 512 + 1024]
    t0 = time.perf_counter()
    rows = ch.process(seg)
    return len(rows) - (lo - h0), time.perf_counter() - t0


def cpu_baseline(seconds_budget: float = 8.0):
    """The oracle (numpy port of dsp/channelizer.py:91-137) timed on the host on a bounded sample, three ways (SURVEY.md
    8(d)): 1 thread (the reference's channelizer is a single-threaded offline loop, benchmark_dsp.py:112-141);
    ThreadPoolExecutor(3) over hop ranges (the pool size the reference's capture uses, capture.py:1918); os.cpu_count()
    worker processes over hop ranges.  `value` / `cores` are the single-thread figure; the others ride along."""
    import numpy as np
    import signals as S
    from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
    from oracle import ref_np as O

    global _CPU_X
    import multiprocessing

    n = 1 << 21
    x = _CPU_X = S.noise_c64(n, 3)
    ch = O.PolyphaseChannelizer(FS, BW)
    ch.process(x[: 1 << 16])
    ch.reset()
    t0 = time.perf_counter()
    reps = 0
    while True:
        ch.process(x)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or reps >= 64:
            break
    msps = reps * n / el / 1e6
    hops = (n - 1024) // 512 + 1
    scaling = [{"cores": 1, "workers": "1 thread", "input_msps": round(msps, 2)}]

    def ranges(k):
        edges = [hops * i // k for i in range(k + 1)]
        return [(edges[i], edges[i + 1], 3, n) for i in range(k) if edges[i + 1] > edges[i]]

    try:
        with ThreadPoolExecutor(3) as ex:                      # the reference's pool size, capture.py:1918
            t1 = time.perf_counter()
            got = sum(r[0] for r in ex.map(_cpu_pfb_worker, ranges(3)))
            el3 = time.perf_counter() - t1
        assert got == hops
        scaling.append({"cores": 3, "workers": "ThreadPoolExecutor(3) over hop ranges", "input_msps": round(n / el3 / 1e6, 2)})
        # all the cores this process may use (the scheduler affinity, not the machine's os.cpu_count(): a one-GPU box is a
        # 16-core share of a 256-thread host), one worker process per core, a hop range of >= 128 hops each
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        nc = max(1, min(avail, hops // 128))
        with ProcessPoolExecutor(nc, mp_context=multiprocessing.get_context("fork")) as ex:   # forked before any GPU use
            list(ex.map(_cpu_pfb_worker, ranges(nc)))          # start the workers (imports) outside the timed pass
            t1 = time.perf_counter()
            got = sum(r[0] for r in ex.map(_cpu_pfb_worker, ranges(nc)))
            elc = time.perf_counter() - t1
        assert got == hops
        scaling.append({"cores": nc, "workers": f"{nc} processes over hop ranges (sched_getaffinity: {avail}, os.cpu_count(): {os.cpu_count()})",
                        "input_msps": round(n / elc / 1e6, 2)})
    except Exception as e:      # the side measurements never take the bench line down
        scaling.append({"error": f"{type(e).__name__}: {e}"})
    return {"value": round(msps * M, 1), "unit": "MS/s x channels", "cores": 1, "kind": "port",
            "sample": f"{reps} x 2^21 complex64 samples through oracle/ref_np.PolyphaseChannelizer "
                      f"(M=1024, numpy {np.__version__}), {el:.1f} s", "input_msps": round(msps, 2),
            "cpu_model": _cpu_model(), "host_cores": os.cpu_count(), "scaling": scaling}


def _profile_counter(kernel_substr: str, counter: str):
    """Average of a hardware counter per launch of a kernel, from the committed rocprofv3 summary of this round
    (the newest profiles/rNN_pmc_bench.json, written by tools/pmc_run.sh + tools/pmc_to_profiles.py); None when absent."""
    import glob
    try:
        d = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_bench.json")))[-1]))
    except Exception:
        return None
    for name, rec in d.items():
        if kernel_substr in name and counter in rec:
            return float(rec[counter])
    return None


def secondary_nbfm(torch, steps: int = 10):
    """BASELINE configs[1]: 32 NBFM channels from one 2.4 MS/s int16-IQ stream, 10 s = 200 chunks of
    120 000 samples per step, one fused launch.  Returns MS/s x channels + the CPU oracle beside it."""
    import numpy as np
    import signals as S
    import wavehip
    from oracle import ref_np as O

    fs, n, K, chunks = 2_400_000, 120_000, 32, 200
    offs = S.nbfm_bank_offsets(K)
    cfgs = [wavehip.ChannelConfig(mode="nbfm", offset_hz=o, enable_deemphasis=False, enable_mpx_filter=False)
            for o in offs]
    bank = wavehip.ChannelBank(fs, n, cfgs, input_format="int16")
    i16_chunk = S.pack_iq16_np(S.nbfm_bank_c64(n, fs, seed=2))
    d_in = torch.from_numpy(np.tile(i16_chunk, chunks)).cuda()
    audio = torch.empty((chunks, K, bank.n_out), dtype=torch.float32, device="cuda")
    met = torch.empty((chunks, K, 4), dtype=torch.float32, device="cuda")
    for _ in range(3):                       # (the first call grows the workspace)
        bank.process_device(d_in, chunks, audio, met)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.process_device(d_in, chunks, audio, met)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    # CPU oracle: 4 channels of one chunk, 1 thread
    z = O.unpack_iq16(i16_chunk)
    t1 = time.perf_counter()
    for k in range(4):
        O.process_channel_nbfm(z, fs, offs[k])
    cpu = 4 * n / (time.perf_counter() - t1) / 1e6
    msps = chunks * n / el / 1e6
    return {"workload": "32x NBFM from one 2.4 MS/s int16 stream, 200 chunks/launch (configs[1])",
            "value": round(msps * K, 1), "unit": "MS/s x channels", "input_msps": round(msps, 1),
            "ms_per_launch": round(el * 1e3, 3), "x_realtime": round(msps / 2.4, 1),
            "algorithmic_GBps": round(6.56 * msps / 1e3, 2),
            **_nbfm_valu(el, chunks * n * K),
            "cpu_port_msps_x_channels": round(cpu, 2), "cpu_cores": 1}


def _nbfm_valu(el: float, sample_channels: float):
    """The second roofline of config 2 ("report it against both"): VALU issue.
    Algorithmic floor: per sample x channel the chain needs the NCO phase (2 ops) + sin and cos (2 transcendentals, quarter
    rate = 8 issue slots) + the mix (1 complex product, 2 packed ops) + the discriminator product (2 packed ops) + atan2
    (1 reciprocal at quarter rate + ~8 polynomial ops = 12 slots) + 1001 taps / 50 = 20 real MACs per input sample
    (10 packed FMAs) = 36 issue slots per lane, i.e. 36 / 64 wave64 instructions; a wave64 instruction holds one of the
    1024 SIMDs for 4 cycles at <= 2.4 GHz.  Issue efficiency: SQ_INSTS_VALU of the shipped kernel, read from this round's
    committed rocprofv3 summary (the newest profiles/rNN_pmc_bench.json), not a constant."""
    slots = 36.0
    floor = sample_channels * slots / 64.0 * 4.0 / 1024.0 / 2.4e9
    out = {"algorithmic_valu_slots_per_sample_channel": slots, "algorithmic_valu_floor_ms": round(floor * 1e3, 3),
           "frac_of_algorithmic_valu_floor": round(floor / el, 4)}
    insts = _profile_counter("fmbank_fused_kernel<false>", "SQ_INSTS_VALU")   # (<true> = the unfused chains' FIR mode)
    if insts:
        issue = insts * 4.0 / 1024.0 / 2.4e9
        out.update({"sq_insts_valu_per_launch": insts, "valu_issue_floor_ms": round(issue * 1e3, 3),
                    "frac_of_valu_issue": round(issue / el, 4)})
    return out


def _secondary_small_rows(torch):
    """Rows whose kernels the verdict asked rocprof evidence for: spectrum (A8), Phase-2 CQPSK and LSM banks (A12).
    Timings only; their kernel statistics come from the rocprofv3 summary of this same command (profiles/)."""
    import wavehip

    out = {}
    N, frames = 2048, 16384
    x = torch.view_as_complex(torch.randn(frames * N, 2, device="cuda") * 0.3)
    be = wavehip.HipFFTBackend(N)
    for _ in range(3):
        be.execute_device(x, frames)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        be.execute_device(x, frames)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 10
    out["spectrum"] = {"workload": "spectrum_mid_kernel (window, 2048-point FFT, |X| dB, fftshift), 16384 frames of complex64 per call "
                                   "(12 B per sample: 8 in, 4 out)",
                       "gsps": round(frames * N / el / 1e9, 1), "algorithmic_GBps": round(12.0 * frames * N / el / 1e9, 1),
                       "frac_of_8TBps": round(12.0 * frames * N / el / 1e9 / HBM_PEAK_GBPS, 4)}
    del x
    for name, mk, fs in (("lsm_bank", lambda: wavehip.LSMBank(64, 19200, 4800, max_samples_per_call=19200), 19200),
                         ("cqpsk_bank", lambda: wavehip.CQPSKBank(64, 48000, 12000, max_samples_per_call=48000), 48000)):
        b = mk()
        z = torch.view_as_complex(torch.randn(64, fs, 2, device="cuda").mul_(0.3))
        for _ in range(2):
            b.demodulate_device(z)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            b.demodulate_device(z)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 3
        out[name] = {"workload": f"64 channels x 1 s at {fs} S/s per call", "ms_per_call": round(el * 1e3, 2),
                     "x_realtime_per_channel": round(1.0 / el, 1)}
    return out


def secondary_pfb_int16(torch, steps: int = 10):
    """The filterbank fed with interleaved int16 IQ (the A1 unpack fused into the loads): 20 algorithmic bytes per
    sample (4 in, 16 out) instead of 24."""
    import wavehip

    n = 1 << 28
    ch = wavehip.PolyphaseChannelizer(FS, BW)
    x = torch.randint(-20000, 20000, (2 * n,), dtype=torch.int16, device="cuda")
    out = torch.empty((ch.hops(n), M), dtype=torch.complex64, device="cuda")
    ch.profile(True)
    ch.plan(x, out)
    for _ in range(20):
        ch.process_device(x, out)
    torch.cuda.synchronize()
    ms = []
    for _ in range(steps):
        ch.process_device(x, out)
        torch.cuda.synchronize()
        ms.append(ch.last_kernel_ms())
    k = sorted(ms)[len(ms) // 2]
    planned = ch.planned
    # yardstick: the same 1 : 4 read : write byte stream with no arithmetic (4 B in, 16 B out per sample)
    from wavehip import _lib as _wl
    del out
    xin = torch.empty(n // 2, 2, dtype=torch.float32, device="cuda").normal_()      # n/2 complex64 = 4 B per sample
    yo = torch.empty(2 * n, dtype=torch.complex64, device="cuda")                    # four copies = 16 B per sample
    for _ in range(3):
        _wl.check(_wl.lib.wh_diag_stream_1r4w(xin.data_ptr(), yo.data_ptr(), n // 2, _wl.stream_ptr(torch)), "diag")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        _wl.check(_wl.lib.wh_diag_stream_1r4w(xin.data_ptr(), yo.data_ptr(), n // 2, _wl.stream_ptr(torch)), "diag")
    torch.cuda.synchronize()
    ys = 20.0 * n / ((time.perf_counter() - t0) / 10) / 1e9
    return {"workload": "1024-channel filterbank, int16 IQ input, 2^28 samples per launch", "kernel_ms": round(k, 4),
            "input_msps": round(n / k / 1e3, 1), "algorithmic_GBps": round(20.0 * n / k / 1e6, 1),
            "frac_of_8TBps": round(20.0 * n / k / 1e6 / 8000.0, 4), "stream_1r4w_yardstick_GBps": round(ys, 1),
            "frac_of_stream_yardstick": round(20.0 * n / k / 1e6 / ys, 4), "planned": planned}


def secondary_pfb_stats(torch, steps: int = 10):
    """The scanner's real workload (A13 / config 5): the 1024-channel filterbank in statistics-only mode -- per-channel
    {sum, sum of squares, min, max} of |y|^2 reduced in the last FFT pass, no channel outputs written: 8 algorithmic bytes
    per input sample (the read).  The arithmetic is unchanged, so this form is bound by VALU / LDS, not by HBM."""
    import wavehip

    n = 1 << 28
    ch = wavehip.PolyphaseChannelizer(FS, BW)
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
    stats = torch.zeros((M, 5), dtype=torch.float64, device="cuda")
    for _ in range(5):
        ch.process_stats_device(x, stats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ch.process_stats_device(x, stats)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    # the same statistics the long way: full output, then wh_pfb_channel_stats over ALL its hops
    out = torch.empty((ch.hops(n), M), dtype=torch.complex64, device="cuda")
    full = wavehip.PolyphaseChannelizer(FS, BW)
    for _ in range(3):
        full.channel_stats_device(full.process_device(x, out), stats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        full.channel_stats_device(full.process_device(x, out), stats)
    torch.cuda.synchronize()
    el2 = (time.perf_counter() - t0) / steps
    return {"workload": "1024-channel filterbank, statistics only (no channel outputs), 2^28 complex64 samples per call",
            "ms_per_call": round(el * 1e3, 4), "input_msps": round(n / el / 1e6, 1),
            "algorithmic_GBps": round(8.0 * n / el / 1e9, 1), "frac_of_8TBps": round(8.0 * n / el / 1e9 / HBM_PEAK_GBPS, 4),
            "ms_full_output_plus_stats_over_all_hops": round(el2 * 1e3, 4), "speedup_vs_that": round(el2 / el, 3),
            "speedup_vs_full_output_pass": None,
            "note": "the tuned 1024-channel kernel in its statistics-only form (round 3): float32 powers meet in LDS per group "
                    "of 4 hops, four channels per thread accumulate float32 blocks into float64; no output buffer at all"}


def secondary_pfb_m320(torch, steps: int = 5):
    """The channelizer shape of the reference's own benchmark_dsp.py:112-141 (8 MS/s, 25 kHz spacing -> M = 320,
    pfb_mid_kernel: 3 runs x 80 quads per 4-wave workgroup, one launch per call) beside the CPU oracle on the same shape."""
    import numpy as np
    import signals as S
    import wavehip
    from oracle import ref_np as O

    fs, bw, n = 8_000_000, 25_000, 1 << 24
    ch = wavehip.PolyphaseChannelizer(fs, bw)
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
    out = torch.empty((ch.hops(n), ch.channel_count), dtype=torch.complex64, device="cuda")
    ch.process_device(x, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ch.process_device(x, out)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ref = O.PolyphaseChannelizer(fs, bw)
    xc = S.noise_c64(1 << 20, 4)
    t0 = time.perf_counter()
    ref.process(xc)
    cpu = (1 << 20) / (time.perf_counter() - t0) / 1e6
    return {"workload": "320-channel filterbank, 8 MS/s cf32 (benchmark_dsp.py shape), 2^24 samples per call",
            "input_msps": round(n / el / 1e6, 1), "x_realtime": round(n / el / fs, 1),
            "algorithmic_GBps": round(24.0 * n / el / 1e9, 1),
            "cpu_port_input_msps": round(cpu, 2), "cpu_cores": 1}


def secondary_wbfm(torch, steps: int = 10):
    """BASELINE configs[0]: ONE default WBFM channel on a 2.4 MS/s complex64 stream, one 120 000-sample chunk per
    call (the live shape: host buffer in, audio out), beside the CPU oracle on the same chunk."""
    import numpy as np
    import signals as S
    import wavehip
    from oracle import ref_np as O

    fs, n = 2_400_000, 120_000
    iq = S.fm_tone_c64(n, fs, seed=3)
    cfg = wavehip.ChannelConfig(mode="wbfm", offset_hz=0.0)
    bank = wavehip.ChannelBank(fs, n, [cfg])
    d_in = torch.from_numpy(iq).cuda()
    bank.process_device(d_in, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.process_device(d_in, 1)
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.process(iq)                          # H2D + launch sequence + D2H + validation
    host = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    O.process_channel_wbfm(iq, fs, 0.0)
    cpu = time.perf_counter() - t0
    return {"workload": "1x WBFM (de-emphasis + MPX low-pass), 2.4 MS/s cf32, one 50 ms chunk per call (configs[0])",
            "ms_per_chunk_device_resident": round(dev * 1e3, 3), "ms_per_chunk_host_to_host": round(host * 1e3, 3),
            "x_realtime": round(0.05 / dev, 1), "cpu_port_ms_per_chunk": round(cpu * 1e3, 2), "cpu_cores": 1}


def secondary_ddc(torch, steps: int = 20):
    """Row N3: the trunking front-end bank -- 64 NCO + two-stage-decimator front-ends (voice-recorder plan,
    trunking/system.py:453-656) on one 100 ms buffer of 6 MS/s IQ per call, beside the CPU port for ONE front-end."""
    import numpy as np
    import signals as S
    import wavehip
    from oracle import ref_np as O

    fs, K = 6_000_000, 64
    n = fs // 10
    x = S.noise_c64(n, 5)
    d = torch.from_numpy(x).cuda()
    bank = wavehip.TrunkingDDCBank(K, fs, plan="recorder", max_samples_per_call=n)
    offs = np.linspace(-2.0e6, 2.0e6, K)
    for _ in range(3):
        bank.process_device(d, offs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.process_device(d, offs)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ref = O.TrunkingDDC(fs, bank.stage1_factor, bank.stage2_factor)
    t0 = time.perf_counter()
    ref.process(x, 250e3)
    cpu = time.perf_counter() - t0
    return {"workload": f"{K} trunking front-ends (NCO + {bank.stage1_factor} x {bank.stage2_factor} decimator) on 100 ms of 6 MS/s cf32 per call",
            "ms_per_call": round(el * 1e3, 3), "x_realtime_for_the_bank": round(0.1 / el, 1),
            "input_gsps_x_channels": round(K * n / el / 1e9, 1),
            "cpu_port_ms_per_call_one_frontend": round(cpu * 1e3, 1), "cpu_cores": 1}


def secondary_c4fm(torch, steps: int = 2):
    """BASELINE configs[3] as SURVEY 8(d) item 4 specifies it: 64 INDEPENDENT P25 C4FM streams at 48 kHz (seeds 1000+k,
    offsets U(-400, 400) Hz, SNR 20 dB), 10 s each, fed in 100 ms calls.  The dibits of all 64 channels are checked
    bit-exact against the C oracle in the same run."""
    import numpy as np
    import wavehip
    from oracle.c4fm_c import C4FMDemodulatorRef
    from signals import config4_streams

    fs, C, call, secs = 48000, 64, 4800, 10
    n = fs * secs
    host, _ = config4_streams(C, fs, n)
    xs = torch.from_numpy(host).cuda()
    bank = wavehip.C4FMBank(C, fs, max_samples_per_call=call)
    got = [[] for _ in range(C)]

    def run(collect):
        bank.reset()
        for s in range(0, n, call):
            d, sf, cnt = bank.demodulate_device(xs[:, s:s + call])
            if collect:
                dc, cc = d.cpu().numpy(), cnt.cpu().numpy()
                for c in range(C):
                    got[c].append(dc[c, :cc[c]].copy())

    run(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(False)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    mism, cpu_s = 0, 0.0
    for c in range(C):
        ref = C4FMDemodulatorRef(sample_rate=fs, atan_mode=1)
        t1 = time.perf_counter()
        rd = np.concatenate([ref.demodulate(host[c, s:s + call])[0] for s in range(0, n, call)])
        cpu_s += time.perf_counter() - t1
        gd = np.concatenate(got[c])
        mism += int(gd.size != rd.size) + int(np.count_nonzero(gd[:rd.size] != rd[:gd.size]))
    sym_s = C * n / (fs / 4800) / el
    # the same streams at the control-channel monitor's call size: 72 000 samples = 1.5 s per demodulate()
    # (trunking/system.py:1548-1549 -> control_channel.py:230-231), 6 calls; dibits of 4 channels against the C oracle
    big, nb = 72000, 6 * 72000
    bank.reserve(big)
    got_b = [[] for _ in range(C)]

    def run_big(collect):
        bank.reset()
        for s in range(0, nb, big):
            d, sf, cnt = bank.demodulate_device(xs[:, s:s + big])
            if collect:
                dc, cc = d.cpu().numpy(), cnt.cpu().numpy()
                for c in range(C):
                    got_b[c].append(dc[c, :cc[c]].copy())

    run_big(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_big(False)
    torch.cuda.synchronize()
    el_b = (time.perf_counter() - t0) / steps
    mism_b = 0
    for c in (0, 21, 42, 63):
        ref = C4FMDemodulatorRef(sample_rate=fs, atan_mode=1)
        rd = np.concatenate([ref.demodulate(host[c, s:s + big])[0] for s in range(0, nb, big)])
        gd = np.concatenate(got_b[c])
        mism_b += int(gd.size != rd.size) + int(np.count_nonzero(gd[:rd.size] != rd[:gd.size]))
    production = {"call_samples": big, "calls": nb // big, "seconds_of_signal": nb / fs, "seconds_per_block": round(el_b, 4),
                  "x_realtime_per_channel": round(nb / fs / el_b, 1), "channels_checked": 4,
                  "dibit_mismatches_vs_c_oracle": mism_b}
    return {"production_call_size": production,
            "workload": "64 independent P25 C4FM streams @48 kHz (seeds 1000+k, offsets U(-400,400) Hz, 20 dB), 10 s, 100 ms calls (configs[3])",
            "symbols_per_s": round(sym_s, 0), "samples_msps_x_channels": round(C * n / el / 1e6, 2),
            "x_realtime_per_channel": round(secs / el, 1), "seconds_per_10s_block": round(el, 4),
            "channels_checked": C, "dibit_mismatches_vs_c_oracle": mism,
            "cpu_port_x_realtime_per_channel": round(C * secs / cpu_s, 1), "cpu_cores": 1}


def drive(step, reducer, steps: int, warmup: int, prewarm: int, sync, barrier, after_step=None):
    """The bench's control flow, shared by every rank: `prewarm` + `warmup` untimed steps, then EXACTLY `steps` timed
    ones bracketed by barrier + synchronize on both sides; the last scan's exchange is collected inside the timed
    region.  Every rank runs the same number of steps, so the one collective per step (reducer.submit inside `step`)
    stays matched across ranks -- tests/test_multi_gpu_cpu.py runs this on two gloo ranks.  Returns (elapsed seconds on
    this rank, merged statistics of the last scan or None)."""
    for _ in range(prewarm):
        step()
    sync()
    for _ in range(warmup):
        step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        if after_step is not None:
            after_step()
    last = reducer.wait(merge=True) if reducer is not None else None
    sync()
    barrier()
    sync()
    return time.perf_counter() - t0, last


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=28, help="samples per step = 2^log2n")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1]/configs[3] side measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # the CPU baseline forks worker processes: it runs first, before this process initialises the GPU
    cpu_line = cpu_baseline() if (world == 1 and not args.no_cpu) else None
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    # WH_BENCH_REHEARSAL=1: control-flow rehearsal of the N>1 path on a 1-GPU box (every rank on
    # cuda:0, gloo instead of RCCL for the tiny stats exchange).  Never set by the driver.
    rehearsal = os.environ.get("WH_BENCH_REHEARSAL") == "1"
    torch.cuda.set_device(0 if rehearsal else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import wavehip
    from wavehip.scanner_reduce import AsyncStatsReducer, reduce_channel_stats

    n = 1 << args.log2n
    ch = wavehip.PolyphaseChannelizer(FS, BW)
    assert ch.channel_count == M
    gen = torch.Generator(device="cuda").manual_seed(3000 + rank)
    x = torch.view_as_complex(torch.randn(n, 2, device="cuda", generator=gen).mul_(0.5))
    hops = ch.hops(n)
    out = torch.empty((hops, M), dtype=torch.complex64, device="cuda")
    stats2 = [torch.zeros((M, 5), dtype=torch.float64, device="cuda") for _ in range(2)]
    scan = min(1024, hops)
    ch.profile(True)
    # planning, outside the timed region (every rank, no collective): the run length a workgroup walks is measured on THIS
    # device for this call size and the fastest kept -- outputs are the same bits for every run length
    ch.plan(x, out)
    reducer = AsyncStatsReducer() if world > 1 else None      # (rehearsal: the same reducer, staged through the host for gloo)
    it = [0]

    def step():
        # filterbank pass + scan-window statistics; for N > 1 the merged view of scan i is collected
        # while pass i+1 runs (one async all-gather per scan, RCCL's own stream)
        stats = stats2[it[0] & 1]
        it[0] += 1
        ch.process_device(x, out)
        ch.channel_stats_device(out[hops - scan:], stats)
        if reducer is not None:
            gathered = reducer.wait(merge=False)   # scan i-1 (stream-ordered wait, no host sync)
            reducer.submit(stats)
            return gathered
        if world > 1:
            return reduce_channel_stats(stats.cpu())
        return stats

    # untimed pre-warm (a FIXED 200 steps ~ 0.3 s, identical on every rank so the collectives stay
    # matched) so the W warmup + K timed steps run at the clocks the chip holds under sustained
    # load rather than on the DVFS ramp of a cold device
    elapsed, last = drive(step, reducer, args.steps, args.warmup, 200, torch.cuda.synchronize,
                          dist.barrier if world > 1 else (lambda: None))
    # HIP events around the fused kernel of every timed step, on the stream it is launched on (the handle keeps the event
    # pairs of its last 64 launches): read AFTER the timed region, so no step waits for the host
    kernel_ms = [ch.last_kernel_ms(back=b) for b in range(min(args.steps, 64))]
    if reducer is not None:
        assert last is not None and last.shape == (M, 5)
        assert reducer.submitted == 200 + args.warmup + args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # outside the timed region: latency of one synchronous scanner/activity exchange
    reduce_ms = None
    if world > 1:
        try:
            st = stats2[0] if not rehearsal else stats2[0].cpu()
            for _ in range(3):
                reduce_channel_stats(st)
            if not rehearsal:
                torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(20):
                reduce_channel_stats(st)
            if not rehearsal:
                torch.cuda.synchronize()
            reduce_ms = (time.perf_counter() - t1) / 20 * 1e3
        except Exception as e:      # never let the side measurement take the bench line down
            print(f"[bench] scanner reduce timing skipped: {e}", file=sys.stderr)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed / 1e6 * M
        k_ms = sum(kernel_ms) / len(kernel_ms)
        fused_samples = ((hops - 8) // 4) * 4 * 512           # samples consumed by the fused kernel's hops
        achieved = BYTES_PER_SAMPLE * fused_samples / (k_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")   # written by tools/pmc_to_profiles.py from this round's passes
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "input IQ MS/s x channels demodulated (1024-channel polyphase filterbank)",
            "value": round(value, 1), "unit": "MS/s x channels", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1024-channel polyphase filterbank, 10 MS/s cf32, channel_bandwidth 9765, "
                                   f"2^{args.log2n} samples per step per GPU, one stream per GPU "
                                   "(BASELINE.json configs[2] / configs[4])",
                       "samples_per_step_per_gpu": n, "hops_per_step": hops, "channels": M,
                       "scan_window_hops": scan, "collective": (f"{dist.get_backend()} all_gather(stats 40 KB/GPU), " + ("async, 1 per step" + (" (gloo rehearsal: staged through the host)" if rehearsal else ""))) if world > 1 else "none",
                       "scanner_reduce_ms": None if reduce_ms is None else round(reduce_ms, 4)},
            "input_msps": round(world * n * args.steps / elapsed / 1e6, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "pfb1024_kernel", "kernel_ms": round(k_ms, 4),
                         "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * fused_samples,
                         "planned": ch.planned},
        }
        if world == 1:
            # yardstick: what a plain device copy of the same read:write mix (1:2) moves on THIS box, same process
            # (HBM efficiency differs by +-5 % between boxes; the guide's ~6.3 TB/s copy figure is a best case)
            src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
            dst = torch.empty(2, 1 << 28, dtype=torch.float32, device="cuda")
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                dst.copy_(src.expand(2, -1))
            ev0.record()
            for _ in range(10):
                dst.copy_(src.expand(2, -1))
            ev1.record()
            torch.cuda.synchronize()
            line["roofline"]["device_copy_yardstick_GBps"] = round(3 * 4 * (1 << 28) * 10 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)
            del src, dst
            # the filterbank's own traffic shape with no arithmetic (hand-written: 16-byte loads, two 16-byte store
            # streams, one pass per workgroup): 2^28 samples in, 2 x 2^28 out -- what the memory system gives this
            # read:write mix when nothing else is in the way
            from wavehip import _lib as _wl
            ys = torch.empty(2 * n + 4096, dtype=torch.complex64, device="cuda")
            for _ in range(3):
                _wl.check(_wl.lib.wh_diag_stream_1r2w(x.data_ptr(), ys.data_ptr(), n, _wl.stream_ptr(torch)), "diag")
            ev0.record()
            for _ in range(10):
                _wl.check(_wl.lib.wh_diag_stream_1r2w(x.data_ptr(), ys.data_ptr(), n, _wl.stream_ptr(torch)), "diag")
            ev1.record()
            torch.cuda.synchronize()
            sy = 24.0 * n * 10 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
            line["roofline"]["stream_1r2w_yardstick_GBps"] = round(sy, 1)
            line["roofline"]["frac_of_stream_yardstick"] = round(line["roofline"]["achieved"] / sy, 4)
            del ys
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        if world == 1 and not args.no_secondary:
            del x, out
            torch.cuda.empty_cache()
            line["secondary"] = {"pfb_int16": secondary_pfb_int16(torch), "pfb_stats_only": secondary_pfb_stats(torch),
                                 "pfb_m320": secondary_pfb_m320(torch),
                                 "wbfm_single": secondary_wbfm(torch),
                                 "nbfm_bank": secondary_nbfm(torch),
                                 "c4fm_bank": secondary_c4fm(torch), "ddc_bank": secondary_ddc(torch),
                                 **_secondary_small_rows(torch)}
        if "secondary" in line and "pfb_stats_only" in line["secondary"]:
            line["secondary"]["pfb_stats_only"]["speedup_vs_full_output_pass"] = round(
                ms_per_step / line["secondary"]["pfb_stats_only"]["ms_per_call"], 3)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
