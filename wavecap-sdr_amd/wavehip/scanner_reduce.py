"""Cross-stream scanner / activity reduction.

Each GPU owns one device stream and produces per-channel BinStats-style statistics
{sum p, sum p^2, count, min p, max p} (channel_classifier.py:17-48); the only cross-GPU step
of the whole path is merging them: sums with all_reduce(SUM), min/max with all_reduce(MIN/MAX)
-- three tiny latency-bound collectives (RCCL over xGMI with backend "nccl", gloo on CPU for
tests).  `best_channel` replicates ControlChannelScanner.get_best_channel's ordering
(cc_scanner.py:355-400: synced first, then SNR descending)."""

from __future__ import annotations


def reduce_channel_stats(stats, group=None):
    """stats: float64 tensor [M, 5] (any device).  Returns the merged tensor (same on all ranks)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return stats
    sums = stats[:, 0:3].contiguous()
    mn = stats[:, 3].contiguous()
    mx = stats[:, 4].contiguous()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return torch.cat([sums, mn[:, None], mx[:, None]], dim=1)


class AsyncStatsReducer:
    """One collective per scan instead of three, overlapped with the next compute step:
    `submit(stats)` copies the packed [M, 5] block into a reducer-owned ping-pong buffer and all-gathers it
    asynchronously (RCCL runs it on its own stream after the producing kernel; the caller may overwrite `stats`
    right away), `wait()` blocks the current stream on it and merges sum / min / max (on the GPU by the
    wh_stats_merge kernel).  One scan can be in flight while the next one is produced; every rank must call
    submit() / wait() the same number of times (each submit is one collective)."""

    def __init__(self, group=None):
        self.group = group
        self._pending = None      # (work, gathered tensor)
        self._send = [None, None]   # ping-pong copies of the submitted block (the collective reads them asynchronously)
        self._recv = [None, None]
        self._k = 0
        self._dev = None
        self.submitted = 0

    def submit(self, stats):
        import torch
        import torch.distributed as dist

        self.wait()
        world = dist.get_world_size(self.group)
        k = self._k
        self._k ^= 1
        # gloo (the one-GPU rehearsal and the CPU tests) gathers host tensors only: device statistics are staged through
        # the host there and come back to the device in wait(), so that the merge kernel runs as it does over RCCL
        self._dev = stats.device
        cdev = torch.device("cpu") if (stats.is_cuda and dist.get_backend(self.group) == "gloo") else stats.device
        if self._send[k] is None or self._send[k].shape != stats.shape or self._send[k].device != cdev:
            self._send[k] = torch.empty(stats.shape, dtype=stats.dtype, device=cdev)
            self._recv[k] = torch.empty((world * stats.shape[0],) + tuple(stats.shape[1:]), dtype=stats.dtype, device=cdev)
        self._send[k].copy_(stats)
        work = dist.all_gather_into_tensor(self._recv[k], self._send[k], group=self.group, async_op=True)
        self._pending = (work, self._recv[k].view((world,) + tuple(stats.shape)))
        self.submitted += 1

    def wait(self, merge: bool = True):
        """Block the CURRENT stream on the pending collective (no host sync).  Returns the merged
        [M, 5] view, or with merge=False the raw gathered [world, M, 5] block (merge it later with
        `merge_gathered`, e.g. once per scan interval instead of once per pass; it aliases a reducer buffer that the
        second next submit() overwrites)."""
        if self._pending is None:
            return None
        work, out = self._pending
        self._pending = None
        work.wait()
        if out.device != self._dev:
            out = out.to(self._dev)
        return self.merge_gathered(out) if merge else out

    @staticmethod
    def merge_gathered(out):
        """[world, M, 5] -> [M, 5]: sums of {sum p, sum p^2, count}, min, max.  On the GPU: the wh_stats_merge kernel
        (fixed rank order, so every rank computes the same bits); host tensors (the gloo tests) are merged here."""
        import torch

        if out.is_cuda and out.dtype == torch.float64 and out.shape[2] == 5:
            from . import _lib

            o = out.contiguous()
            res = torch.empty(o.shape[1:], dtype=torch.float64, device=o.device)
            _lib.check(_lib.lib.wh_stats_merge(o.data_ptr(), o.shape[0], o.shape[1], res.data_ptr(), _lib.stream_ptr(torch)),
                       "wh_stats_merge")
            return res
        # same convention as the kernel: a rank whose count is 0 contributes no min / max; nothing anywhere: +inf / -inf
        has = out[:, :, 2] > 0
        inf = torch.tensor(float("inf"), dtype=out.dtype, device=out.device)
        mn = torch.where(has, out[:, :, 3], inf).min(0).values
        mx = torch.where(has, out[:, :, 4], -inf).max(0).values
        return torch.cat([out[:, :, 0:3].sum(0), mn[:, None], mx[:, None]], dim=1)


def gather_measurements(meas, group=None):
    """meas: float64 tensor [n, F] of per-stream candidate measurements
    (power_db, snr_db, sync flag ...).  Returns [world, n, F] on every rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return meas[None]
    out = [torch.empty_like(meas) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, meas.contiguous(), group=group)
    return torch.stack(out)


def activity_from_stats(stats, squelch_db: float):
    """ScannerService._is_activity_detected (scanner.py:203-208): rssi > squelch, with
    rssi = 10 log10(mean power + 1e-10) per channel."""
    import torch

    mean_p = stats[:, 0] / torch.clamp(stats[:, 2], min=1.0)
    rssi = 10.0 * torch.log10(mean_p + 1e-10)
    return rssi > squelch_db, rssi


def best_channel(snr_db, synced):
    """cc_scanner.py:377-400: prefer synced candidates, then highest SNR.  Returns the flat index."""
    import torch

    score = snr_db.clone().double()
    score = torch.where(synced.bool(), score + 1e6, score)
    return int(torch.argmax(score.reshape(-1)))
