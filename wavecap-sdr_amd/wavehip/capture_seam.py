"""The batched seam: drop-in for Capture._process_channels_parallel (capture.py:2489-2597, SURVEY.md 8(b) B2 batched
form / 8(f) N1).  Same signature and conventions --

    process_channels_parallel(capture, samples, executor, timeout=0.5) -> list[(Channel, audio | None)]

so that `Capture._process_channels_parallel = wavehip.process_channels_parallel` (a module-level rebind, the injection
the reference's own tests use, tests/unit/test_capture_dsp_timeout.py:28) makes a capture run every running channel of a
chunk through ONE upload and one launch sequence per chain group instead of one thread-pool task per channel:

  * running channels only, in `capture._channels` order;
  * back-pressure: the capture's `_dsp_inflight` counter / lock are honoured (a chunk counts as one job in flight; when
    the backlog is at its limit the cycle is skipped and [] returned, as the reference does);
  * the chunk's DSP runs on the caller's executor, so the capture thread stays responsive and the wait can time out:
    on timeout every channel gets (channel, None) -- the reference's convention for a late channel -- and the job is
    cancelled if it has not started;
  * per channel afterwards, exactly like the reference: rssi_db / signal_power_db stored on the channel,
    `capture._apply_stateful_processing(ch, audio, samples)` (RDS / POCSAG / squelch), `ch._update_audio_metrics(audio)`;
    an exception in that tail gives (channel, None) and a log line, never propagates.
The stateful decoders and everything after the tuple list stay the reference's own code.
"""

from __future__ import annotations

import logging
import time
from concurrent.futures import wait
from typing import Any

logger = logging.getLogger(__name__)


def dispatch_chunk(capture, samples, cfgs) -> list[tuple[Any, dict[str, Any]]]:
    """The chunk's DSP for all channels: [(audio | None, metrics)] in cfg order (ChannelDispatcher, cached on the
    capture).  A module attribute so that tests / integrators can rebind it, like the reference's
    _process_channel_dsp_stateless."""
    from .channel_ops import ChannelDispatcher

    disp = getattr(capture, "_wavehip_dispatcher", None)
    if disp is None or disp.sample_rate != int(capture.cfg.sample_rate):
        disp = ChannelDispatcher(int(capture.cfg.sample_rate))
        capture._wavehip_dispatcher = disp
    return disp.process(samples, cfgs)


def process_channels_parallel(capture, samples, executor, timeout: float = 0.5) -> list[tuple[Any, Any]]:
    channels = [ch for ch in capture._channels.values() if ch.state == "running"]
    if not channels:
        return []
    max_workers = getattr(executor, "_max_workers", 4)
    max_inflight = max(4, max_workers * 2)
    lock = capture._dsp_inflight_lock
    with lock:
        if capture._dsp_inflight >= max_inflight:
            now = time.time()
            if now - getattr(capture, "_dsp_drop_last_log", 0.0) >= 2.0:
                logger.warning(f"Capture {capture.cfg.id}: DSP backlog {capture._dsp_inflight}/{max_inflight}, "
                               "skipping cycle to recover")
                capture._dsp_drop_last_log = now
            return []
        capture._dsp_inflight += 1

    # the module attribute is looked up at call time so that a rebind (tests, integrators) takes effect
    import wavehip.capture_seam as seam

    future = executor.submit(seam.dispatch_chunk, capture, samples, [ch.cfg for ch in channels])

    def _done(_fut) -> None:
        with lock:
            capture._dsp_inflight = max(0, capture._dsp_inflight - 1)

    future.add_done_callback(_done)
    done, not_done = wait([future], timeout=timeout)
    if not_done:
        for ch in channels:
            logger.warning(f"Channel {ch.cfg.id} DSP timeout (batched wait)")
        future.cancel()
        return [(ch, None) for ch in channels]
    try:
        per_channel = future.result(timeout=0)
    except Exception as e:  # noqa: BLE001  (the reference logs and drops the audio of a failed channel)
        for ch in channels:
            logger.error(f"Channel {ch.cfg.id} DSP error: {e}")
        return [(ch, None) for ch in channels]
    results: list[tuple[Any, Any]] = []
    for ch, (audio, metrics) in zip(channels, per_channel):
        try:
            if "rssi_db" in metrics:
                ch.rssi_db = metrics["rssi_db"]
            if "signal_power_db" in metrics:
                ch.signal_power_db = metrics["signal_power_db"]
            audio = capture._apply_stateful_processing(ch, audio, samples)
            if audio is not None and audio.size > 0:
                ch._update_audio_metrics(audio)
            results.append((ch, audio))
        except Exception as e:  # noqa: BLE001
            logger.error(f"Channel {ch.cfg.id} DSP error: {e}")
            results.append((ch, None))
    return results
