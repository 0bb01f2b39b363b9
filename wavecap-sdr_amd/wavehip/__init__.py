"""wavehip -- MI355X-native implementation of WaveCap-SDR's per-channel DSP hot path.

Host-side mirror of the reference's operator surfaces (same names, arguments and
error behaviour) over libwavehip.so (hand-written HIP for gfx950):

    PolyphaseChannelizer      <- wavecapsdr/dsp/channelizer.py:28
    HipFFTBackend             <- wavecapsdr/dsp/fft/base.py:31 (FFTBackend plugin)
    process_channel_dsp_stateless / ChannelBank  <- wavecapsdr/capture.py:298
    C4FMDemodulator / C4FMBank <- wavecapsdr/dsp/p25/c4fm.py:2379

PyTorch-ROCm is used for device buffers and streams only.  There is no CPU
fallback: without the built library (or without a GPU) the operators raise.
"""

from ._lib import LIB_PATH, lib  # noqa: F401  (raises ImportError if the .so is missing)
from .channelizer import PolyphaseChannelizer, ChannelCalculator  # noqa: F401
from .fft_backend import HipFFTBackend, FFTResult, FFTBackend, is_available, register_with  # noqa: F401
from .channel_ops import (ChannelBank, ChannelConfig, ChannelDispatcher, process_channel_dsp_stateless,  # noqa: F401
                          update_signal_metrics, noise_blanker)
from .capture_seam import process_channels_parallel  # noqa: F401
from . import channel_split  # noqa: F401
from .wire import pack_iq16, unpack_iq16, pack_pcm16, pack_f32  # noqa: F401
from .framer import (P25P1SoftSyncDetector, SoftSyncBank, P25NIDFrontEnd, NACTracker,  # noqa: F401
                     strip_status_symbols, strip_status_symbols_device)
from .fec import BCHDecoder, bch_decode  # noqa: F401
from .c4fm import C4FMBank, C4FMDemodulator, c4fm_demod_simple  # noqa: F401
from .cqpsk import (CQPSKBank, CQPSKDemodulator, GardnerBank, GardnerTED, CostasBank, CostasLoop,  # noqa: F401
                    MuellerMullerBank, MuellerMullerTED)
from .lsm import LSMBank, LSMDemodulator  # noqa: F401
from .classifier import ChannelClassifier, ClassifiedChannel  # noqa: F401
from .trunking import (TrunkingDDC, TrunkingDDCBank, ScannerMeasure, decimation_plan,  # noqa: F401
                       recorder_decimation_plan)

__all__ = [
    "PolyphaseChannelizer", "ChannelCalculator", "HipFFTBackend", "FFTResult", "FFTBackend", "is_available",
    "register_with", "process_channels_parallel", "ChannelBank", "ChannelConfig", "ChannelDispatcher", "noise_blanker", "process_channel_dsp_stateless", "update_signal_metrics", "pack_iq16", "unpack_iq16",
    "pack_pcm16", "pack_f32", "P25P1SoftSyncDetector", "SoftSyncBank", "P25NIDFrontEnd", "NACTracker", "strip_status_symbols", "strip_status_symbols_device", "BCHDecoder", "bch_decode", "C4FMBank", "C4FMDemodulator", "c4fm_demod_simple", "CQPSKBank", "CQPSKDemodulator", "GardnerBank", "GardnerTED", "CostasBank", "CostasLoop", "MuellerMullerBank", "MuellerMullerTED", "LSMBank", "LSMDemodulator", "ChannelClassifier", "ClassifiedChannel", "TrunkingDDC", "TrunkingDDCBank", "ScannerMeasure", "decimation_plan", "recorder_decimation_plan",
]
