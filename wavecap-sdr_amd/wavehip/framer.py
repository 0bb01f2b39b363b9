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


class NACTracker:
    """decoders/nac_tracker.py: up to 3 NACs with observation counts; the dominant one (>= 3 observations) assists
    the BCH second pass; when a 4th NAC appears the least recently seen is dropped."""

    MAX_TRACKER_COUNT = 3
    MIN_OBSERVATION_THRESHOLD = 3

    def __init__(self) -> None:
        self._count: dict[int, int] = {}
        self._seen: dict[int, int] = {}
        self._tick = 0

    def reset(self) -> None:
        self._count.clear()
        self._seen.clear()

    def track(self, nac: int) -> None:
        if nac < 0 or nac > 0xFFF:
            return
        self._tick += 1
        if nac in self._count:
            self._count[nac] += 1
            self._seen[nac] = self._tick
        else:
            self._count[nac] = 1
            self._seen[nac] = self._tick
            if len(self._count) > self.MAX_TRACKER_COUNT:
                oldest = min(self._seen, key=lambda k: self._seen[k])
                del self._count[oldest], self._seen[oldest]

    def get_tracked_nac(self) -> int:
        if not self._count:
            return 0
        nac = max(self._count, key=lambda k: self._count[k])       # first maximum in insertion order, like max()
        return nac if self._count[nac] >= self.MIN_OBSERVATION_THRESHOLD else 0


def strip_status_symbols_device(dibits_dev, initial_counter: int = 21):
    """uint8 GPU tensor [n] or [C, n] -> the rows without their status symbols (uint8 GPU tensor [kept] / [C, kept])."""
    import ctypes as C

    torch = _lib.require_gpu()
    assert dibits_dev.is_cuda and dibits_dev.dtype == torch.uint8 and dibits_dev.dim() in (1, 2)
    x = dibits_dev if dibits_dev.dim() == 2 else dibits_dev[None, :]
    assert x.shape[1] == 0 or x.stride(1) == 1
    rows, n = x.shape
    out = torch.empty((rows, n), dtype=torch.uint8, device=x.device)
    kept = C.c_size_t(0)
    _lib.check(_lib.lib.wh_strip_status(x.data_ptr() if n else None, n, x.stride(0) if rows > 1 else max(n, 1), rows,
                                        int(initial_counter), out.data_ptr() if n else None, max(n, 1), C.byref(kept),
                                        _lib.stream_ptr(torch)), "wh_strip_status")
    out = out[:, :kept.value]
    return out if dibits_dev.dim() == 2 else out[0]


def strip_status_symbols(dibits, initial_counter: int = 21) -> np.ndarray:
    """Drop-in for P25Decoder._strip_status_symbols (decoders/p25.py:2816-2862): raw dibits with a status symbol every 36
    dibits of the frame -> the clean dibit stream (uint8).  `initial_counter` is the frame position of the first dibit
    modulo 36 (21 for a TSDU starting at frame position 57)."""
    d = np.ascontiguousarray(np.asarray(dibits), dtype=np.uint8)
    if d.size == 0:
        return np.array([], dtype=np.uint8)
    torch = _lib.require_gpu()
    return strip_status_symbols_device(torch.from_numpy(d).cuda(), initial_counter).cpu().numpy()


class P25NIDFrontEnd:
    """Front half of P25P1MessageFramer.process_batch (p25_framer.py:471-617) on the device: soft sync scores ->
    positions above SYNC_DETECTION_THRESHOLD -> the 33 dibits collected from each position (a later sync within the
    33 restarts the collection) -> status dibit stripped -> BCH(63,16,23) with the tracked-NAC second pass.
    `process_batch(soft_symbols, dibits)` returns the NID events [(index of the completing dibit in the stream since
    reset, nac, duid, bit_errors)]; what happens after a valid NID (message assembly, trellis, TSBK) is protocol
    logic downstream of this path.  In batch mode the reference starts collecting AT the dibit that completes the
    sync (the callback runs before that dibit is processed, p25_framer.py:503-509) -- kept."""

    DIBIT_LENGTH_NID = 33
    SYNC_DETECTION_THRESHOLD = 60.0

    def __init__(self) -> None:
        from .fec import BCHDecoder
        self._torch = _lib.require_gpu()
        self._sync = SoftSyncBank(1)
        self._bch = BCHDecoder()
        self.nac_tracker = NACTracker()
        self._pos_dev = self._cnt_dev = None
        self.reset()

    def reset(self) -> None:
        self._sync.reset()
        self.nac_tracker.reset()
        self._tail = np.zeros(0, dtype=np.uint8)     # dibits of a collection still open at the end of the last call
        self._tail_start = 0                         # stream index of _tail[0]
        self._pos = 0                                # stream index of the next dibit

    def process_batch(self, soft_symbols, dibits) -> list[tuple[int, int, int, int]]:
        torch = self._torch
        dib = np.ascontiguousarray(dibits, dtype=np.uint8)
        soft = np.ascontiguousarray(soft_symbols, dtype=np.float32)
        n = dib.size
        if n == 0 or soft.size != n:
            return []
        scores = self._sync.process_device(torch.from_numpy(soft[None, :]).cuda())
        if self._pos_dev is None or self._pos_dev.numel() < n:      # work buffers live with the object (grow only)
            self._pos_dev = torch.empty(max(n, 4096), dtype=torch.int32, device="cuda")
            self._cnt_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
        pos_dev, cnt_dev = self._pos_dev, self._cnt_dev              # (wh_sync_positions zeroes the counter itself)
        _lib.check(_lib.lib.wh_sync_positions(scores.data_ptr(), n, float(self.SYNC_DETECTION_THRESHOLD),
                                              pos_dev.data_ptr(), n, cnt_dev.data_ptr(), _lib.stream_ptr(torch)),
                   "wh_sync_positions")
        cnt = int(cnt_dev.item())
        new = np.sort(pos_dev[:cnt].cpu().numpy()).astype(np.int64) + self._pos
        # stream = carried tail + this call
        base = self._tail_start if self._tail.size else self._pos
        stream = np.concatenate([self._tail, dib]) if self._tail.size else dib
        starts = ([self._tail_start] if self._tail.size else []) + [int(p) for p in new]
        end = self._pos + n
        # a later sync before the 33rd dibit restarts the collection
        live = [s for i, s in enumerate(starts) if i + 1 == len(starts) or starts[i + 1] > s + self.DIBIT_LENGTH_NID - 1]
        done = [s for s in live if s + self.DIBIT_LENGTH_NID <= end]
        open_ = [s for s in live if s + self.DIBIT_LENGTH_NID > end]
        events: list[tuple[int, int, int, int]] = []
        if done:
            d_stream = torch.from_numpy(np.ascontiguousarray(stream)).cuda()
            d_starts = torch.from_numpy(np.array([s - base for s in done], dtype=np.int32)).cuda()
            words = torch.empty(len(done), dtype=torch.int64, device="cuda")
            _lib.check(_lib.lib.wh_nid_extract(d_stream.data_ptr(), stream.size, d_starts.data_ptr(), len(done),
                                               words.data_ptr(), _lib.stream_ptr(torch)), "wh_nid_extract")
            data, err = (t.cpu().numpy() for t in self._bch.decode_device(words))
            second: dict[int, tuple] = {}            # tracked NAC -> second-pass results of all words
            for k, s in enumerate(done):
                dat, e = int(data[k]), int(err[k])
                tracked = self.nac_tracker.get_tracked_nac()
                if e < 0 and tracked:
                    if tracked not in second:
                        tr = torch.full((len(done),), tracked, dtype=torch.int32, device="cuda")
                        second[tracked] = tuple(t.cpu().numpy() for t in self._bch.decode_device(words, tr))
                    dat, e = int(second[tracked][0][k]), int(second[tracked][1][k])
                if e < 0:
                    continue
                nac, duid = (dat >> 4) & 0xFFF, dat & 0xF
                self.nac_tracker.track(nac)
                events.append((s + self.DIBIT_LENGTH_NID - 1, nac, duid, e))
        if open_:
            s = open_[-1]
            self._tail = stream[s - base:].copy()
            self._tail_start = s
        else:
            self._tail = np.zeros(0, dtype=np.uint8)
        self._pos = end
        return events
