"""P25 Phase-1 soft sync detector on the MI355X: drop-in for
wavecapsdr.decoders.p25_framer.P25P1SoftSyncDetector (p25_framer.py:124-231) -- `process(soft)`,
`process_batch(soft_symbols) -> scores float32`, `reset()` -- and a bank over channels."""

from __future__ import annotations

import numpy as np

from . import _lib

SYNC_PATTERN = 0x5575F5FF77FF


class SoftSyncBank:
    def __init__(self, n_channels: int):
        self._torch = _lib.require_gpu()
        self.n_channels = int(n_channels)
        torch = self._torch
        self._hist = [torch.zeros((self.n_channels, 24), dtype=torch.float32, device="cuda") for _ in range(2)]
        self._cur = 0

    def reset(self) -> None:
        for h in self._hist:
            h.zero_()

    def process_device(self, soft_dev):
        """soft_dev: float32 GPU tensor [C, n] -> scores float32 [C, n] (score i = after symbol i)."""
        torch = self._torch
        assert soft_dev.is_cuda and soft_dev.dtype == torch.float32 and soft_dev.dim() == 2
        assert soft_dev.shape[0] == self.n_channels and soft_dev.stride(1) == 1
        n = soft_dev.shape[1]
        scores = torch.empty((self.n_channels, n), dtype=torch.float32, device=soft_dev.device)
        if n == 0:
            return scores
        stride = soft_dev.stride(0) if self.n_channels > 1 else n
        assert scores.stride(0) == n and (self.n_channels == 1 or stride == n), "rows must be dense"
        _lib.check(_lib.lib.wh_sync_correlate(soft_dev.data_ptr(), n, stride, self.n_channels,
                                              self._hist[self._cur].data_ptr(), self._hist[self._cur ^ 1].data_ptr(),
                                              scores.data_ptr(), _lib.stream_ptr(torch)), "wh_sync_correlate")
        self._cur ^= 1
        return scores


class P25P1SoftSyncDetector:
    SYNC_PATTERN = SYNC_PATTERN

    def __init__(self) -> None:
        self.SYNC_PATTERN_SYMBOLS = np.array(
            [3.0 if ((SYNC_PATTERN >> ((23 - i) * 2)) & 3) == 1 else -3.0 for i in range(24)], dtype=np.float32)
        self._bank = SoftSyncBank(1)

    def reset(self) -> None:
        self._bank.reset()

    def process_batch(self, soft_symbols) -> np.ndarray:
        x = np.ascontiguousarray(soft_symbols, dtype=np.float32)
        if x.size == 0:
            return np.array([], dtype=np.float32)
        torch = self._bank._torch
        return self._bank.process_device(torch.from_numpy(x[None, :]).cuda())[0].cpu().numpy()

    def process(self, soft_symbol: float) -> float:
        return float(self.process_batch(np.array([soft_symbol], dtype=np.float32))[0])
