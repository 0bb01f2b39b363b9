"""FFT backend plugin "hip" for the reference's spectrum registry.

Mirrors wavecapsdr/dsp/fft/base.py:17-81 (FFTResult / FFTBackend) and the backend
contract of dsp/fft/scipy_backend.py:38-79: `execute(iq, sample_rate)` uses the first
`fft_size` samples, returns fftshifted power in dB (float32), the frequency axis
(float32) and bin_hz; short input -> zero arrays, not an exception.  Construction
raises ImportError when no GPU / library is present so that the reference registry
(dsp/fft/registry.py:41-53) falls through to scipy exactly as it does for cupy.

When the reference package is importable its own base classes are used, so an
instance passes `isinstance(x, wavecapsdr.dsp.fft.base.FFTBackend)`.
"""

from __future__ import annotations

import ctypes as C
from abc import ABC, abstractmethod
from dataclasses import dataclass

import numpy as np

try:  # drop-in: subclass the reference's own ABC when it is there
    from wavecapsdr.dsp.fft.base import FFTBackend, FFTResult  # type: ignore
except Exception:  # standalone mirror (same fields / same abstract surface)

    @dataclass
    class FFTResult:  # base.py:17-28
        power_db: np.ndarray
        freqs: np.ndarray
        bin_hz: float

    class FFTBackend(ABC):  # base.py:31-81
        def __init__(self, fft_size: int = 2048):
            self.fft_size = fft_size
            self._window = None

        @property
        def window(self) -> np.ndarray:
            if self._window is None or len(self._window) != self.fft_size:
                self._window = np.hanning(self.fft_size).astype(np.float32)
            return self._window

        @abstractmethod
        def execute(self, iq, sample_rate: int) -> "FFTResult":
            ...

        @property
        @abstractmethod
        def name(self) -> str:
            ...

        def __repr__(self) -> str:
            return f"{self.__class__.__name__}(fft_size={self.fft_size})"


def is_available() -> bool:
    """Module-level probe, like dsp/fft/cupy_backend.py:127-129."""
    try:
        from . import _lib
        import torch

        return bool(torch.cuda.is_available()) and _lib.lib is not None
    except Exception:
        return False


class HipFFTBackend(FFTBackend):
    """MI355X spectrum backend (window + FFT + |.| + fftshift + log10 fused in one HIP kernel)."""

    def __init__(self, fft_size: int = 2048, engine: str = "fused"):
        """engine "fused": one HIP kernel (window, LDS FFT, |.|, shift, log); the library picks the shaped kernel for
        fft_size 256 .. 4096, the Stockham / direct-DFT kernel otherwise.  "stockham" / "shaped" force one of the two
        (tests, measurements).
        engine "rocfft": HIP window kernel -> rocFFT batched C2C (reached through torch.fft, which is
        hipFFT/rocFFT on ROCm) -> HIP epilogue kernel; the form BASELINE.json's north_star names."""
        if engine not in ("fused", "rocfft", "stockham", "shaped"):
            raise ValueError("engine must be 'fused', 'stockham', 'shaped' or 'rocfft'")
        self.engine = engine
        try:
            from . import _lib
            import torch
        except Exception as e:  # missing .so or torch
            raise ImportError(f"wavehip HIP backend unavailable: {e}") from e
        if not torch.cuda.is_available():
            raise ImportError("wavehip HIP backend unavailable: no ROCm GPU")
        super().__init__(fft_size)
        self._lib, self._torch = _lib, torch
        self._h = C.c_void_p()
        self._destroy = _lib.lib.wh_spectrum_destroy
        _lib.check(_lib.lib.wh_spectrum_create(C.byref(self._h), int(fft_size)), "wh_spectrum_create")
        if engine in ("stockham", "shaped"):
            _lib.check(_lib.lib.wh_spectrum_tune(self._h, 1, 1 if engine == "stockham" else 2), "wh_spectrum_tune")
        self._freq_cache: dict[int, np.ndarray] = {}

    def __del__(self):
        h, destroy = getattr(self, "_h", None), getattr(self, "_destroy", None)
        if h and destroy:
            destroy(h)
            self._h = None

    def _freqs(self, sample_rate: int) -> np.ndarray:
        f = self._freq_cache.get(sample_rate)
        if f is None:
            f = np.fft.fftshift(np.fft.fftfreq(self.fft_size, 1.0 / sample_rate)).astype(np.float32)
            self._freq_cache = {sample_rate: f}
        return f

    def execute_device(self, iq_dev, n_frames: int = 1, frame_stride: int | None = None):
        """Batched: complex64 GPU tensor -> float32 GPU tensor [n_frames, fft_size]."""
        torch = self._torch
        N = self.fft_size
        stride = N if frame_stride is None else int(frame_stride)
        assert iq_dev.is_cuda and iq_dev.dtype == torch.complex64 and iq_dev.is_contiguous()
        assert iq_dev.numel() >= (n_frames - 1) * stride + N
        out = torch.empty((n_frames, N), dtype=torch.float32, device=iq_dev.device)
        if self.engine == "rocfft":
            w = torch.empty((n_frames, N), dtype=torch.complex64, device=iq_dev.device)
            self._lib.check(self._lib.lib.wh_spectrum_window(self._h, iq_dev.data_ptr(), n_frames, stride, w.data_ptr(),
                                                             self._lib.stream_ptr(torch)), "wh_spectrum_window")
            X = torch.fft.fft(w, dim=1)          # rocFFT, batched C2C forward
            self._lib.check(self._lib.lib.wh_spectrum_post(self._h, X.data_ptr(), n_frames, out.data_ptr(),
                                                           self._lib.stream_ptr(torch)), "wh_spectrum_post")
            return out
        self._lib.check(self._lib.lib.wh_spectrum_run(self._h, iq_dev.data_ptr(), n_frames, stride, out.data_ptr(),
                                                      self._lib.stream_ptr(torch)), "wh_spectrum_run")
        return out

    def execute(self, iq, sample_rate: int) -> FFTResult:
        N = self.fft_size
        if iq.size < N:  # scipy_backend.py:48-54
            return FFTResult(power_db=np.zeros(N, dtype=np.float32), freqs=np.zeros(N, dtype=np.float32),
                             bin_hz=sample_rate / N)
        torch = self._torch
        chunk = np.ascontiguousarray(iq[:N], dtype=np.complex64)
        p = self.execute_device(torch.from_numpy(chunk).cuda(), 1)[0].cpu().numpy()
        return FFTResult(power_db=p, freqs=self._freqs(sample_rate).copy(), bin_hz=sample_rate / N)

    @property
    def name(self) -> str:
        return "hip"


def register_with(registry_module) -> None:
    """Register under the name "hip" in the reference registry (dsp/fft/registry.py:24-38):
    `register_with(wavecapsdr.dsp.fft.registry)`; then CaptureConfig.fft_accelerator="hip"."""
    registry_module._ensure_registered()
    if is_available():
        registry_module._BACKENDS["hip"] = HipFFTBackend
