"""Trunking front-end operators on the MI355X.

`TrunkingDDC`  -- the control-channel IQ path of trunking/system.py:1735-1779: phase-continuous
                  NCO (system.py:1434-1466) + two-stage FIR decimation (system.py:1318-1406,
                  dsp/filters.py:613-646), computing kept outputs only.
`ScannerMeasure` -- ControlChannelScanner._measure_channel (trunking/cc_scanner.py:165-264):
                  per-candidate power / peak / edge-noise / SNR, all candidates in one launch.
Filter designs are the reference's own scipy calls."""

from __future__ import annotations

import ctypes as C

import numpy as np
from scipy import signal

from . import _lib


def decimation_plan(input_rate: int, target_output_rate: int = 48000) -> tuple[int, int]:
    """Stage factors exactly as trunking/system.py:1310-1345."""
    total = max(1, input_rate // target_output_rate)
    s1f, s2f = total, 1
    if total >= 48:
        for s1 in [30, 24, 20, 16, 12, 10, 8]:
            if total % s1 == 0 and 2 <= total // s1 <= 10:
                s1f, s2f = s1, total // s1
                break
        else:
            s1f = int(total ** 0.5)
            s2f = total // s1f
    else:
        for s1 in [12, 10, 8, 6, 4]:
            if total % s1 == 0 and 2 <= total // s1 <= 10:
                s1f, s2f = s1, total // s1
                break
        else:
            s1f, s2f = total, 1
    return s1f, s2f


def recorder_decimation_plan(sample_rate: int, target_rate: int = 48000) -> tuple[int, int]:
    """VoiceRecorder.setup_decimation_filter (trunking/system.py:453-485): two stages only from 100:1 up."""
    total = max(1, sample_rate // target_rate)
    if total <= 1:
        return 1, 1
    if total >= 100:
        for s1 in [25, 20, 30, 16]:
            if total % s1 == 0 and total // s1 <= 10:
                return s1, total // s1
    return total, 1


class TrunkingDDC:
    def __init__(self, sample_rate: int, stage1_factor: int | None = None, stage2_factor: int | None = None,
                 max_samples_per_call: int = 1 << 20):
        self._torch = _lib.require_gpu()
        self.sample_rate = int(sample_rate)
        if stage1_factor is None:
            stage1_factor, stage2_factor = decimation_plan(self.sample_rate)
        self.stage1_factor, self.stage2_factor = int(stage1_factor), int(stage2_factor or 1)
        # system.py:1392-1406
        self.stage1_taps = signal.firwin(157, 0.8 / self.stage1_factor, window=("kaiser", 7.857))
        self.stage2_taps = (signal.firwin(73, 0.8 / self.stage2_factor, window=("kaiser", 7.857))
                            if self.stage2_factor > 1 else np.zeros(1))
        self.output_rate = self.sample_rate // self.stage1_factor // self.stage2_factor
        t1 = np.ascontiguousarray(self.stage1_taps, dtype=np.float64)
        t2 = np.ascontiguousarray(self.stage2_taps, dtype=np.float64)
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_ddc_destroy
        _lib.check(_lib.lib.wh_ddc_create(C.byref(self._h), self.sample_rate, _lib.dptr(t1, "f64"), len(t1),
                                          self.stage1_factor, _lib.dptr(t2, "f64"), len(t2), self.stage2_factor,
                                          int(max_samples_per_call)), "wh_ddc_create")

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self) -> None:
        _lib.check(_lib.lib.wh_ddc_reset(self._h), "wh_ddc_reset")

    def out_len(self, n: int) -> int:
        return int(_lib.lib.wh_ddc_out_len(self._h, n))

    def process_device(self, iq_dev, offset_hz: float):
        torch = self._torch
        assert iq_dev.is_cuda and iq_dev.dtype == torch.complex64 and iq_dev.is_contiguous()
        n = iq_dev.numel()
        out = torch.empty(self.out_len(n), dtype=torch.complex64, device=iq_dev.device)
        _lib.check(_lib.lib.wh_ddc_run(self._h, iq_dev.data_ptr(), n, float(offset_hz), out.data_ptr(),
                                       _lib.stream_ptr(torch)), "wh_ddc_run")
        return out

    def process(self, iq, offset_hz: float) -> np.ndarray:
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        if x.size == 0:
            return x
        return self.process_device(self._torch.from_numpy(x).cuda(), offset_hz).cpu().numpy()


class TrunkingDDCBank:
    """n_channels front-ends (NCO + two-stage decimator, each with its own offset, phase index and filter state)
    on one wideband buffer per call: the voice-recorder pool of trunking/system.py:453-656 plus the control
    monitor.  plan="recorder" uses VoiceRecorder.setup_decimation_filter's stage factors, plan="control" the
    control monitor's (system.py:1310-1345); explicit factors override."""

    MAX_CHANNELS = 64

    def __init__(self, n_channels: int, sample_rate: int, plan: str = "recorder", stage1_factor: int | None = None,
                 stage2_factor: int | None = None, max_samples_per_call: int = 1 << 20):
        self._torch = _lib.require_gpu()
        self.n_channels, self.sample_rate = int(n_channels), int(sample_rate)
        if stage1_factor is None:
            stage1_factor, stage2_factor = (recorder_decimation_plan if plan == "recorder" else decimation_plan)(
                self.sample_rate)
        self.stage1_factor, self.stage2_factor = int(stage1_factor), int(stage2_factor or 1)
        if self.stage1_factor < 2:
            raise ValueError("TrunkingDDCBank needs a decimating first stage")
        self.stage1_taps = signal.firwin(157, 0.8 / self.stage1_factor, window=("kaiser", 7.857))
        self.stage2_taps = (signal.firwin(73, 0.8 / self.stage2_factor, window=("kaiser", 7.857))
                            if self.stage2_factor > 1 else np.zeros(2))
        self.output_rate = self.sample_rate // self.stage1_factor // self.stage2_factor
        t1 = np.ascontiguousarray(self.stage1_taps, dtype=np.float64)
        t2 = np.ascontiguousarray(self.stage2_taps, dtype=np.float64)
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_ddc_bank_destroy
        _lib.check(_lib.lib.wh_ddc_bank_create(C.byref(self._h), self.n_channels, self.sample_rate, _lib.dptr(t1, "f64"),
                                               len(t1), self.stage1_factor, _lib.dptr(t2, "f64"), len(t2),
                                               self.stage2_factor, int(max_samples_per_call)), "wh_ddc_bank_create")

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def reset(self, channel: int = -1) -> None:
        _lib.check(_lib.lib.wh_ddc_bank_reset(self._h, int(channel)), "wh_ddc_bank_reset")

    def out_len(self, n: int) -> int:
        return int(_lib.lib.wh_ddc_bank_out_len(self._h, n))

    def process_device(self, iq_dev, offsets_hz, active=None):
        """iq_dev complex64 GPU tensor [n] -> complex64 [n_channels, out_len(n)]; rows of inactive channels are
        left untouched (uninitialised on first use)."""
        torch = self._torch
        assert iq_dev.is_cuda and iq_dev.dtype == torch.complex64 and iq_dev.is_contiguous()
        n = iq_dev.numel()
        m = self.out_len(n)
        out = torch.empty((self.n_channels, m), dtype=torch.complex64, device=iq_dev.device)
        offs = np.ascontiguousarray(offsets_hz, dtype=np.float64)
        assert offs.shape == (self.n_channels,)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        _lib.check(_lib.lib.wh_ddc_bank_run(self._h, iq_dev.data_ptr(), n, _lib.dptr(offs, "f64"),
                                            None if act is None else act.ctypes.data, out.data_ptr(), m,
                                            _lib.stream_ptr(torch)), "wh_ddc_bank_run")
        return out

    def process(self, iq, offsets_hz, active=None) -> list:
        """-> per channel: complex64 array, or None for an inactive channel."""
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        if x.size == 0:
            return [x for _ in range(self.n_channels)]
        out = self.process_device(self._torch.from_numpy(x).cuda(), offsets_hz, active).cpu().numpy()
        return [out[k] if active is None or active[k] else None for k in range(self.n_channels)]


class ScannerMeasure:
    """cc_scanner.py:165-264 for a list of candidate offsets of one wideband buffer."""

    def __init__(self, sample_rate: int, sync_check_enabled: bool = True):
        self._torch = _lib.require_gpu()
        self.sample_rate = int(sample_rate)
        self.sync_check_enabled = sync_check_enabled
        self.decim = max(1, self.sample_rate // 48000)
        self.taps = np.ascontiguousarray(
            signal.firwin(65, 0.8 / self.decim, window=("kaiser", 6.0)) if self.decim > 1 else np.ones(1),
            dtype=np.float64)

    def power(self, iq_dev, offsets_hz, want_sync: bool = False):
        """-> float64 [n, 2] = (mean |y|^2, max |y|^2) per offset (and the best sync correlation [n])."""
        torch = self._torch
        assert iq_dev.is_cuda and iq_dev.dtype == torch.complex64 and iq_dev.is_contiguous()
        offs = np.array([int(round(float(o))) if float(o) != 0.0 else 0 for o in offsets_hz], dtype=np.int32)
        out = np.zeros((len(offs), 2), dtype=np.float64)
        corr = np.zeros(len(offs), dtype=np.float64)
        _lib.check(_lib.lib.wh_scan_measure(iq_dev.data_ptr(), iq_dev.numel(), self.sample_rate, _lib.dptr(offs, "i32"),
                                            len(offs), _lib.dptr(self.taps, "f64"), len(self.taps), self.decim,
                                            out.ctypes.data_as(C.POINTER(C.c_double)),
                                            corr.ctypes.data_as(C.POINTER(C.c_double)) if want_sync else None,
                                            _lib.stream_ptr(torch)), "wh_scan_measure")
        return (out, corr) if want_sync else out

    def measure(self, iq, channel_offsets_hz) -> list[dict]:
        """Per candidate: power_db, peak_power_db, noise_floor_db, snr_db, sync_detected, sample_count -- the
        fields of ChannelMeasurement (cc_scanner.py:240-264)."""
        torch = self._torch
        x = iq if torch.is_tensor(iq) else torch.from_numpy(np.ascontiguousarray(iq, dtype=np.complex64)).cuda()
        max_offset = self.sample_rate / 2 - 15000
        edges = [-max_offset + 25000, max_offset - 25000]
        p, corr = self.power(x, list(channel_offsets_hz) + edges, want_sync=True)
        noise = min(p[-2, 0], p[-1, 0])
        eps = 1e-12
        out = []
        for i in range(len(channel_offsets_hz)):
            pw = 10 * np.log10(p[i, 0] + eps)
            nf = 10 * np.log10(noise + eps)
            snr = float(pw - nf)
            sync = bool(self.sync_check_enabled and snr >= 8.0 and abs(corr[i]) > 0.6)   # cc_scanner.py:247-251, 345-350
            out.append(dict(power_db=float(pw), peak_power_db=float(10 * np.log10(p[i, 1] + eps)),
                            noise_floor_db=float(nf), snr_db=snr, sync_detected=sync, sync_correlation=float(corr[i]),
                            sample_count=(x.numel() + self.decim - 1) // self.decim))
        return out
