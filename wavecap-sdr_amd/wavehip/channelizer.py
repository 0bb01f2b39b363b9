"""Drop-in for wavecapsdr.dsp.channelizer.PolyphaseChannelizer (channelizer.py:28-158).

Same constructor, attributes (`channel_count`, `channel_sample_rate`, `arms`,
`arm_history`), `process`, `reset`, `extract_channel`.  `process()` returns one
complex64 array of shape [hops, channel_count]; iterating it / indexing it / len()
behave like the reference's list of per-hop vectors, and `extract_channel` accepts
either form.  `process_device()` keeps input and output in HBM.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
from scipy import signal

from . import _lib

DEFAULT_CHANNEL_BANDWIDTH = 25000
DEFAULT_TAPS_PER_CHANNEL = 9


class PolyphaseChannelizer:
    def __init__(self, sample_rate: float, channel_bandwidth: int = DEFAULT_CHANNEL_BANDWIDTH,
                 taps_per_channel: int = DEFAULT_TAPS_PER_CHANNEL):
        self.sample_rate = sample_rate
        self.channel_bandwidth = channel_bandwidth
        self.taps_per_channel = taps_per_channel
        # channelizer.py:52-58
        self.channel_count = int(sample_rate / channel_bandwidth)
        if self.channel_count % 2 != 0:
            self.channel_count -= 1
        self.channel_sample_rate = (sample_rate / self.channel_count) * 2
        self._design_filter()
        self.block_counter = 0
        self._torch = _lib.require_gpu()
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_pfb_destroy
        arms = np.ascontiguousarray(self.arms, dtype=np.float64)
        _lib.check(_lib.lib.wh_pfb_create(C.byref(self._h), self.channel_count, self.taps_per_channel,
                                          _lib.dptr(arms, "f64")), "wh_pfb_create")

    def _design_filter(self) -> None:
        """channelizer.py:69-89 (same scipy call -> identical prototype taps)."""
        M, T = self.channel_count, self.taps_per_channel
        cutoff = (self.channel_bandwidth * 0.9) / (self.sample_rate / 2)
        proto = signal.firwin(M * T - 1, cutoff, window=("kaiser", 8.0)).astype(np.float64)
        self.arms = np.zeros((M, T), dtype=np.float64)
        for arm in range(M):
            t = proto[arm::M]
            self.arms[arm, : len(t)] = t

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    # -- carried state ---------------------------------------------------------------------
    @property
    def arm_history(self) -> np.ndarray:
        out = np.empty((self.channel_count, self.taps_per_channel), dtype=np.complex64)
        _lib.check(_lib.lib.wh_pfb_get_history(self._h, out.ctypes.data, _lib.stream_ptr(self._torch)),
                   "wh_pfb_get_history")
        return out

    @arm_history.setter
    def arm_history(self, value) -> None:
        v = np.ascontiguousarray(value, dtype=np.complex64)
        assert v.shape == (self.channel_count, self.taps_per_channel)
        _lib.check(_lib.lib.wh_pfb_set_history(self._h, v.ctypes.data, _lib.stream_ptr(self._torch)),
                   "wh_pfb_set_history")

    def reset(self) -> None:
        """channelizer.py:139-142."""
        _lib.check(_lib.lib.wh_pfb_reset(self._h, _lib.stream_ptr(self._torch)), "wh_pfb_reset")
        self.block_counter = 0

    # -- processing ------------------------------------------------------------------------
    def hops(self, n_samples: int) -> int:
        return int(_lib.lib.wh_pfb_hops(self._h, n_samples))

    def process_device(self, samples, out=None):
        """samples: complex64 (or interleaved int16 IQ) torch tensor on the GPU -> complex64 tensor
        [hops, M] on the GPU."""
        torch = self._torch
        assert samples.is_cuda and samples.is_contiguous()
        if samples.dtype == torch.int16:          # interleaved int16 IQ: unpack fused into the loads
            n, run = samples.numel() // 2, _lib.lib.wh_pfb_run_i16
        else:
            assert samples.dtype == torch.complex64
            n, run = samples.numel(), _lib.lib.wh_pfb_run
        H = self.hops(n)
        if out is None:
            out = torch.empty((H, self.channel_count), dtype=torch.complex64, device=samples.device)
        else:
            assert out.is_cuda and out.dtype == torch.complex64 and out.is_contiguous()
            assert out.numel() >= H * self.channel_count
        _lib.check(run(self._h, samples.data_ptr(), n, out.data_ptr(), _lib.stream_ptr(torch)), "wh_pfb_run")
        return out

    def process(self, samples) -> np.ndarray:
        """channelizer.py:91-137.  Host array in, host array [hops, M] out."""
        torch = self._torch
        x = np.ascontiguousarray(samples, dtype=np.complex64)
        if self.hops(x.shape[0]) == 0:
            return np.zeros((0, self.channel_count), dtype=np.complex64)
        d = torch.from_numpy(x).cuda()
        return self.process_device(d).cpu().numpy()

    TUNE_KEYS = {"path": 1, "prefetch": 2, "hops_per_run": 3, "run_map": 5, "alt_dir": 6}
    PATHS = {"auto": 0, "per_hop": 1, "run": 2, "shaped": 3}

    def tune(self, **kw) -> "PolyphaseChannelizer":
        """Explicit kernel selection for tests / measurements (wh_pfb_tune): path="auto"|"per_hop"|"run"|"shaped",
        prefetch=0|1|3|5|7 (5 / 7: the three-workgroups-per-CU forms), hops_per_run=int, run_map=0|-1|-2|C, alt_dir=0|1."""
        for k, v in kw.items():
            if k == "path" and isinstance(v, str):
                v = self.PATHS[v]
            _lib.check(_lib.lib.wh_pfb_tune(self._h, self.TUNE_KEYS[k], int(v)), "wh_pfb_tune")
        return self

    def plan(self, samples, out=None, candidates=None, rounds: int = 8) -> int:
        """Measure-and-pick planning for calls of this size, on THIS device (the FFTW-planner idea): how many hops a
        workgroup walks decides how the launch's read and write streams meet the HBM channels, and the best length
        differs between otherwise identical MI355X boxes by more than any kernel change of rounds 2-3 moved the time
        (tools/ubench/stream_shapes.hip: the same no-arithmetic walk moves 4.9-6.3 TB/s depending on run length, mapping
        and resident workgroups; tools/pfb_gpw_sweep.py: one box is fastest at 32 hops, the next at 256).  Every
        candidate is timed on the caller's buffers by HIP events in interleaved rounds; the fastest median is set with
        tune(hops_per_run=...) and returned (0 = the built-in choice).  Outputs and carried history are unaffected (every
        run length gives the same bits); the history is restored afterwards."""
        import statistics

        M1024 = self.channel_count == 1024 and self.taps_per_channel == 9
        if candidates is None:      # 1024 channels: groups of 4 hops per workgroup; shaped kernels: hops per run
            candidates = (0, 3, 4, 5, 6, 8, 12, 64) if M1024 else (0, 16, 32, 64, 128)
        hist = self.arm_history
        times: dict[int, list[float]] = {c: [] for c in candidates}
        self.profile(True)
        for r in range(rounds + 1):
            for c in candidates:
                self.tune(hops_per_run=c)
                out = self.process_device(samples, out)
                if r:                                     # the first round warms every variant up
                    times[c].append(self.last_kernel_ms())
        best = min(candidates, key=lambda c: statistics.median(times[c]))
        self.tune(hops_per_run=best)
        self.arm_history = hist
        self.planned = {"hops_per_run_setting": best, "median_ms": {int(c): round(statistics.median(t), 4) for c, t in times.items()}}
        return best

    def profile(self, enable: bool = True) -> None:
        _lib.check(_lib.lib.wh_pfb_profile(self._h, 1 if enable else 0), "wh_pfb_profile")

    def last_kernel_ms(self, back: int = 0) -> float:
        """Duration of the main filterbank kernel of the most recent profiled launch (back = 0) or of the launch `back`
        launches earlier (the handle keeps 64): waits for that launch only."""
        ms = C.c_float()
        _lib.check(_lib.lib.wh_pfb_kernel_ms_back(self._h, int(back), C.byref(ms)), "wh_pfb_kernel_ms_back")
        return float(ms.value)

    def channel_stats_device(self, out_dev, stats=None, accumulate: bool = False):
        """A13 activity statistics of a filterbank output block: float64 [M, 5] on the GPU."""
        torch = self._torch
        H = out_dev.shape[0]
        if stats is None:
            stats = torch.zeros((self.channel_count, 5), dtype=torch.float64, device=out_dev.device)
            accumulate = False
        _lib.check(_lib.lib.wh_pfb_channel_stats(self._h, out_dev.data_ptr(), H, stats.data_ptr(),
                                                 1 if accumulate else 0, _lib.stream_ptr(torch)),
                   "wh_pfb_channel_stats")
        return stats

    def process_stats_device(self, samples, stats=None, accumulate: bool = False):
        """Statistics-only pass (wh_pfb_run_stats): the per-channel activity statistics [M, 5] float64 of ALL hops of
        `samples` (GPU tensor, complex64 or interleaved int16) without writing the channel outputs; history carried like
        process_device()."""
        torch = self._torch
        assert samples.is_cuda and samples.is_contiguous()
        fmt = 1 if samples.dtype == torch.int16 else 0
        n = samples.numel() // (2 if fmt else 1)
        if stats is None:
            stats = torch.zeros((self.channel_count, 5), dtype=torch.float64, device=samples.device)
            accumulate = False
        _lib.check(_lib.lib.wh_pfb_run_stats(self._h, samples.data_ptr(), fmt, n, stats.data_ptr(), 1 if accumulate else 0,
                                             _lib.stream_ptr(torch)), "wh_pfb_run_stats")
        return stats

    def extract_channel(self, channel_results, channel_index: int):
        """channelizer.py:144-158."""
        torch = self._torch
        if torch.is_tensor(channel_results) and channel_results.is_cuda:
            H = channel_results.shape[0]
            col = torch.empty(H, dtype=torch.complex64, device=channel_results.device)
            _lib.check(_lib.lib.wh_pfb_extract_channel(channel_results.data_ptr(), H, self.channel_count,
                                                       int(channel_index), col.data_ptr(), _lib.stream_ptr(torch)),
                       "wh_pfb_extract_channel")
            return col
        return np.array([r[channel_index] for r in channel_results], dtype=np.complex64)


class ChannelCalculator:
    """channelizer.py:161-231 (pure index arithmetic, host side)."""

    def __init__(self, center_frequency: float, sample_rate: float,
                 channel_bandwidth: int = DEFAULT_CHANNEL_BANDWIDTH):
        self.center_frequency = center_frequency
        self.sample_rate = sample_rate
        self.channel_bandwidth = channel_bandwidth
        self.channel_count = int(sample_rate / channel_bandwidth)
        if self.channel_count % 2 != 0:
            self.channel_count -= 1

    def get_channel_index(self, target_frequency: float) -> int:
        off = int(round((target_frequency - self.center_frequency) / self.channel_bandwidth))
        return self.channel_count + off if off < 0 else off % self.channel_count

    def get_channel_center_frequency(self, channel_index: int) -> float:
        if channel_index < self.channel_count // 2:
            return self.center_frequency + channel_index * self.channel_bandwidth
        return self.center_frequency + (channel_index - self.channel_count) * self.channel_bandwidth
