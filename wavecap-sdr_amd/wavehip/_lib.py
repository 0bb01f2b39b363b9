"""ctypes binding of libwavehip.so (C ABI declared in include/wavehip.h).

The library is the product: importing this module raises ImportError when the
shared object is missing, and every wrapper raises RuntimeError on a non-zero
status -- there is no CPU fallback anywhere in this package.
"""

from __future__ import annotations

import ctypes as C
import os

# torch first: its wheel bundles the HIP runtime (SONAME libamdhip64.so.7); loading it before
# libwavehip.so makes both share ONE runtime instance, so torch's device pointers and streams
# are valid inside the library.  (Loaded the other way round, /opt/rocm's copy wins and torch
# no longer finds the device.)
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwavehip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C wavecap-sdr_amd` (or "
        "`python -c 'import __graft_entry__ as g; g.build()'`); wavehip has no CPU fallback"
    )

lib = C.CDLL(LIB_PATH)

c_void_p, c_int, c_size_t, c_float, c_double = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double


class IirStage(C.Structure):
    _fields_ = [("is_f64", c_int), ("n", c_int), ("b", c_double * 11), ("a", c_double * 11)]


class ChanBankCfg(C.Structure):
    _fields_ = [
        ("sample_rate", c_int),
        ("chunk_len", c_int),
        ("n_channels", c_int),
        ("h_offsets_hz", C.POINTER(c_int)),
        ("input_format", c_int),
        ("demod", c_int),
        ("bfo_hz", c_double),
        ("n_stages", c_int),
        ("h_stages", C.POINTER(IirStage)),
        ("agc", c_int),
        ("agc_target", c_float),
        ("agc_max_gain", c_float),
        ("agc_att_b0", c_float),
        ("agc_att_a1", c_float),
        ("agc_rel_b0", c_float),
        ("agc_rel_a1", c_float),
        ("post", c_int),
        ("h_taps", C.POINTER(c_double)),
        ("ntaps", c_int),
        ("up", c_int),
        ("down", c_int),
        ("d0", c_int),
        ("n_out", c_int),
        ("pll_alpha", c_double),
        ("pll_beta", c_double),
        ("noise_reduction", c_int),
        ("nr_reduction_linear", c_float),
        ("h_nr_window", C.POINTER(c_float)),
        ("iir_warmup", c_int),
        ("iir_scan", c_int),
        ("iir_warmup_form", c_int),
        ("h_squelch_db", C.POINTER(c_float)),
    ]


# name -> (restype, argtypes).  Must list every symbol include/wavehip.h declares
# (tests/test_abi.py parses the header and checks both directions).
PROTOTYPES = {
    "wh_abi_version": (c_int, []),
    "wh_last_error": (C.c_char_p, []),
    "wh_device_info": (c_int, [C.POINTER(c_int), C.POINTER(c_int), C.c_char_p, c_size_t]),
    "wh_unpack_i16_cf32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wh_pack_cf32_i16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wh_pack_f32_pcm16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wh_clip_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wh_sync_correlate": (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wh_sync_positions": (c_int, [c_void_p, c_size_t, c_float, c_void_p, c_size_t, c_void_p, c_void_p]),
    "wh_nid_extract": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "wh_strip_status": (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p, c_size_t, C.POINTER(c_size_t), c_void_p]),
    "wh_bch_create": (c_int, [C.POINTER(c_void_p), c_void_p]),
    "wh_bch_decode": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wh_bch_destroy": (None, [c_void_p]),
    "wh_binstats_update": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    "wh_noise_blanker": (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_int, c_void_p]),
    "wh_audio_stats": (c_int, [c_void_p, c_size_t, C.POINTER(c_float), c_void_p]),
    "wh_nco_mix": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "wh_fm_discriminate": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "wh_resampler_create": (c_int, [C.POINTER(c_void_p), C.POINTER(c_double), c_int, c_int, c_int, c_int]),
    "wh_resampler_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p]),
    "wh_resampler_destroy": (None, [c_void_p]),
    "wh_chanbank_create": (c_int, [C.POINTER(c_void_p), C.POINTER(ChanBankCfg)]),
    "wh_chanbank_run": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "wh_chanbank_run_wire": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "wh_chanbank_set_offsets": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "wh_chanbank_set_squelch": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "wh_chanbank_workspace_bytes": (c_size_t, [c_void_p, c_size_t]),
    "wh_chanbank_destroy": (None, [c_void_p]),
    "wh_channel_signal_metrics": (c_int, [c_void_p, c_int, c_size_t, c_int, C.POINTER(c_int), c_int, C.POINTER(c_float),
                                          c_void_p]),
    "wh_pfb_create": (c_int, [C.POINTER(c_void_p), c_int, c_int, C.POINTER(c_double)]),
    "wh_pfb_hops": (c_size_t, [c_void_p, c_size_t]),
    "wh_pfb_run": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "wh_pfb_run_i16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "wh_pfb_reset": (c_int, [c_void_p, c_void_p]),
    "wh_pfb_get_history": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wh_pfb_set_history": (c_int, [c_void_p, c_void_p, c_void_p]),
    "wh_costas_bank_create": (c_int, [C.POINTER(c_void_p), c_int, c_double, c_double, c_double]),
    "wh_costas_bank_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),
    "wh_costas_bank_reset": (c_int, [c_void_p, c_void_p]),
    "wh_costas_bank_destroy": (None, [c_void_p]),
    "wh_mm_bank_create": (c_int, [C.POINTER(c_void_p), c_int, c_double, c_double, c_double]),
    "wh_mm_bank_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "wh_mm_bank_reset": (c_int, [c_void_p, c_void_p]),
    "wh_mm_bank_destroy": (None, [c_void_p]),
    "wh_pfb_run_stats": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_int, c_void_p]),
    "wh_stats_merge": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "wh_pfb_tune": (c_int, [c_void_p, c_int, c_int]),
    "wh_pfb_profile": (c_int, [c_void_p, c_int]),
    "wh_pfb_kernel_ms": (c_int, [c_void_p, C.POINTER(c_float)]),
    "wh_pfb_kernel_ms_back": (c_int, [c_void_p, c_int, C.POINTER(c_float)]),
    "wh_pfb_extract_channel": (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    "wh_pfb_channel_stats": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "wh_diag_stream_1r2w": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wh_diag_stream_1r4w": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "wh_pfb_destroy": (None, [c_void_p]),
    "wh_spectrum_create": (c_int, [C.POINTER(c_void_p), c_int]),
    "wh_spectrum_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "wh_spectrum_tune": (c_int, [c_void_p, c_int, c_int]),
    "wh_spectrum_window": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "wh_spectrum_post": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "wh_spectrum_destroy": (None, [c_void_p]),
    "wh_ddc_bank_create": (c_int, [C.POINTER(c_void_p), c_int, c_int, C.POINTER(c_double), c_int, c_int,
                                   C.POINTER(c_double), c_int, c_int, c_int]),
    "wh_ddc_bank_out_len": (c_size_t, [c_void_p, c_size_t]),
    "wh_ddc_bank_run": (c_int, [c_void_p, c_void_p, c_size_t, C.POINTER(c_double), c_void_p, c_void_p, c_size_t, c_void_p]),
    "wh_ddc_bank_reset": (c_int, [c_void_p, c_int]),
    "wh_ddc_bank_destroy": (None, [c_void_p]),
    "wh_ddc_create": (c_int, [C.POINTER(c_void_p), c_int, C.POINTER(c_double), c_int, c_int, C.POINTER(c_double),
                              c_int, c_int, c_int]),
    "wh_ddc_out_len": (c_size_t, [c_void_p, c_size_t]),
    "wh_ddc_run": (c_int, [c_void_p, c_void_p, c_size_t, c_double, c_void_p, c_void_p]),
    "wh_ddc_reset": (c_int, [c_void_p]),
    "wh_ddc_destroy": (None, [c_void_p]),
    "wh_scan_measure": (c_int, [c_void_p, c_size_t, c_int, C.POINTER(c_int), c_int, C.POINTER(c_double), c_int, c_int,
                                C.POINTER(c_double), C.POINTER(c_double), c_void_p]),
    "wh_lsm_bank_create": (c_int, [C.POINTER(c_void_p), c_int, c_double, C.POINTER(c_float), C.POINTER(c_float), c_int]),
    "wh_lsm_bank_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p,
                                c_void_p]),
    "wh_lsm_bank_reserve": (c_int, [c_void_p, c_int, c_void_p]),
    "wh_lsm_bank_reset": (c_int, [c_void_p, c_void_p]),
    "wh_lsm_bank_get_state": (c_int, [c_void_p, c_int, C.POINTER(c_double), c_void_p]),
    "wh_lsm_bank_destroy": (None, [c_void_p]),
    "wh_cqpsk_bank_create": (c_int, [C.POINTER(c_void_p), c_int, c_double, C.POINTER(c_float), c_int,
                                     C.POINTER(c_double), c_double, c_double, c_double, c_double, c_double, c_int]),
    "wh_cqpsk_bank_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p,
                                  c_void_p]),
    "wh_cqpsk_bank_reserve": (c_int, [c_void_p, c_size_t, c_void_p]),
    "wh_cqpsk_bank_reset": (c_int, [c_void_p, c_void_p]),
    "wh_cqpsk_bank_destroy": (None, [c_void_p]),
    "wh_gardner_bank_create": (c_int, [C.POINTER(c_void_p), c_int, c_double, c_double, c_double]),
    "wh_gardner_bank_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p,
                                    c_void_p]),
    "wh_gardner_bank_reset": (c_int, [c_void_p, c_void_p]),
    "wh_gardner_bank_destroy": (None, [c_void_p]),
    "wh_c4fm_bank_create": (c_int, [C.POINTER(c_void_p), c_int, c_double, C.POINTER(c_float), c_int,
                                    C.POINTER(c_float), c_int, C.POINTER(c_float), c_int]),
    "wh_c4fm_bank_run": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_size_t,
                                 c_void_p, c_void_p]),
    "wh_c4fm_bank_reserve": (c_int, [c_void_p, c_size_t, c_void_p]),
    "wh_c4fm_bank_reset": (c_int, [c_void_p, c_void_p]),
    "wh_c4fm_bank_destroy": (None, [c_void_p]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)  # AttributeError here == library/headers out of sync: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib.wh_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libwavehip {what} failed ({status}): {msg}")


def dptr(arr, dtype: str) -> C.POINTER:
    """Host numpy array -> typed ctypes pointer (keeps no reference: caller holds the array)."""
    import numpy as np

    ct = {"f64": c_double, "f32": c_float, "i32": c_int}[dtype]
    npdt = {"f64": np.float64, "f32": np.float32, "i32": np.int32}[dtype]
    assert arr.dtype == npdt and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.POINTER(ct))


def require_gpu():
    """torch (ROCm) is used for device buffers and streams only."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("wavehip needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback exists")
    return torch


def stream_ptr(torch) -> int:
    return int(torch.cuda.current_stream().cuda_stream)
