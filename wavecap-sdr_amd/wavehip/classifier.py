"""Spectrum channel classifier with device-resident bin statistics: drop-in for
wavecapsdr.channel_classifier.ChannelClassifier (channel_classifier.py:62-238).  `update()` folds spectrum
frames -- a Python list, a numpy array or a GPU tensor straight from `HipFFTBackend`, one frame [N] or a batch
[F, N] -- into per-bin {sum, sum_sq, count, min, max} on the device (the reference's per-bin Python loop and the
`.tolist()` before it are the spectrum path's real cost); `classify()` reads the 5 N doubles back at most once a
second and runs the reference's peak picking on them."""

from __future__ import annotations

import math
import threading
import time
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _lib


@dataclass
class ClassifiedChannel:
    freq_hz: float
    power_db: float
    std_dev_db: float
    channel_type: str          # "control" | "voice" | "variable" | "unknown"


class ChannelClassifier:
    def __init__(self, min_collection_seconds: float = 60.0, min_samples_per_bin: int = 50,
                 noise_threshold_db: float = -50.0, control_variance_threshold: float = 4.0,
                 voice_variance_threshold: float = 10.0):
        self._torch = _lib.require_gpu()
        self.min_collection_seconds = min_collection_seconds
        self.min_samples_per_bin = min_samples_per_bin
        self.noise_threshold_db = noise_threshold_db
        self.control_variance_threshold = control_variance_threshold
        self.voice_variance_threshold = voice_variance_threshold
        self._stats = None            # GPU float64 [N, 5]
        self._sample_count = 0
        self._start_time: float | None = None
        self._center_hz = 0.0
        self._sample_rate = 0.0
        self._freqs: Any = []
        self._lock = threading.Lock()
        self._cached_channels: list[ClassifiedChannel] = []
        self._last_classify_time = 0.0

    def _clear(self) -> None:
        self._stats = None
        self._sample_count = 0
        self._start_time = None
        self._cached_channels = []

    def reset(self) -> None:
        with self._lock:
            self._clear()
            self._last_classify_time = 0.0

    def update(self, power_db, freqs, center_hz: float, sample_rate: float) -> None:
        torch = self._torch
        with self._lock:
            if center_hz != self._center_hz or sample_rate != self._sample_rate:
                self._clear()
                self._center_hz = center_hz
                self._sample_rate = sample_rate
            self._freqs = freqs
            if self._start_time is None:
                self._start_time = time.time()
            if torch.is_tensor(power_db):
                p = power_db.to(device="cuda", dtype=torch.float32)
            else:
                p = torch.from_numpy(np.ascontiguousarray(power_db, dtype=np.float32)).cuda()
            if p.dim() == 1:
                p = p[None, :]
            p = p.contiguous()
            frames, n = p.shape
            if self._stats is not None and self._stats.shape[0] < n:      # a longer frame adds new bins
                grown = self._new_stats(n)
                grown[:self._stats.shape[0]] = self._stats
                self._stats = grown
            if self._stats is None:
                self._stats = self._new_stats(n)
            if self._stats.shape[0] != n:                                 # shorter frame: only its bins are updated
                sub = self._stats[:n].contiguous()
                _lib.check(_lib.lib.wh_binstats_update(p.data_ptr(), frames, n, sub.data_ptr(), _lib.stream_ptr(torch)),
                           "wh_binstats_update")
                self._stats[:n] = sub
            else:
                _lib.check(_lib.lib.wh_binstats_update(p.data_ptr(), frames, n, self._stats.data_ptr(),
                                                       _lib.stream_ptr(torch)), "wh_binstats_update")
            self._sample_count += frames

    def _new_stats(self, n: int):
        torch = self._torch
        s = torch.zeros((n, 5), dtype=torch.float64, device="cuda")
        s[:, 3] = math.inf
        s[:, 4] = -math.inf
        return s

    def bin_stats(self) -> np.ndarray:
        """float64 [N, 5] = sum, sum_sq, count, min, max per bin (host copy)."""
        return np.zeros((0, 5)) if self._stats is None else self._stats.cpu().numpy()

    @property
    def elapsed_seconds(self) -> float:
        return 0.0 if self._start_time is None else time.time() - self._start_time

    @property
    def is_ready(self) -> bool:
        return self.elapsed_seconds >= self.min_collection_seconds

    @property
    def sample_count(self) -> int:
        return self._sample_count

    def classify(self, force: bool = False) -> list[ClassifiedChannel]:
        now = time.time()
        with self._lock:
            if not force and now - self._last_classify_time < 1.0 and self._cached_channels:
                return list(self._cached_channels)
            if not self.is_ready or self._stats is None:
                return []
            st = self._stats.cpu().numpy()
            total, sq, cnt = st[:, 0], st[:, 1], st[:, 2]
            with np.errstate(divide="ignore", invalid="ignore"):
                mean = np.where(cnt > 0, total / np.maximum(cnt, 1), 0.0)
                var = np.where(cnt < 2, 0.0, sq / np.maximum(cnt, 1) - mean * mean)
            std = np.sqrt(np.maximum(0.0, var))
            ok = cnt >= self.min_samples_per_bin
            if not ok.any():
                return []
            averages = np.sort(mean[ok])
            noise_floor = float(averages[int(len(averages) * 0.2)])
            signal_threshold = noise_floor + 10
            n = len(mean)
            order = [int(i) for i in np.nonzero(ok)[0]]
            order.sort(key=lambda i: mean[i], reverse=True)          # stable, like sorted(..., reverse=True)
            visited: set[int] = set()
            out: list[ClassifiedChannel] = []
            for i in order:
                if i in visited:
                    continue
                avg = float(mean[i])
                if avg < signal_threshold:
                    continue
                prev_avg = float(mean[i - 1]) if i - 1 >= 0 else -math.inf
                next_avg = float(mean[i + 1]) if i + 1 < n else -math.inf
                if avg <= prev_avg or avg <= next_avg:
                    continue
                for off in range(-3, 4):
                    visited.add(i + off)
                freq_hz = self._center_hz + (float(self._freqs[i]) if i < len(self._freqs) else 0)
                sd = float(std[i])
                if avg < noise_floor + 5:
                    kind = "unknown"
                elif sd < self.control_variance_threshold:
                    kind = "control"
                elif sd > self.voice_variance_threshold:
                    kind = "voice"
                else:
                    kind = "variable"
                out.append(ClassifiedChannel(freq_hz=freq_hz, power_db=avg, std_dev_db=sd, channel_type=kind))
            out.sort(key=lambda c: c.power_db, reverse=True)
            self._cached_channels = out
            self._last_classify_time = now
            return list(out)

    def get_status(self) -> dict[str, Any]:
        return {
            "elapsed_seconds": round(self.elapsed_seconds, 1),
            "sample_count": self._sample_count,
            "is_ready": self.is_ready,
            "remaining_seconds": max(0, round(self.min_collection_seconds - self.elapsed_seconds, 1)),
            "center_hz": self._center_hz,
            "sample_rate": self._sample_rate,
        }
