"""The C-ABI library loads on a CPU-only host and exports exactly what include/wavehip.h
declares (no compute calls here)."""

import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "wavehip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(wh_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_exported():
    from wavehip import _lib

    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in wavehip.h but not exported by libwavehip.so"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes PROTOTYPES and wavehip.h are out of sync"


def test_version_and_error_text():
    from wavehip import _lib

    assert _lib.lib.wh_abi_version() == 1
    assert isinstance(_lib.lib.wh_last_error(), bytes)


def test_argument_errors_do_not_need_a_gpu():
    from wavehip import _lib

    h = ctypes.c_void_p()
    rc = _lib.lib.wh_pfb_create(ctypes.byref(h), 7, 9, None)   # odd M, null taps
    assert rc == -1 and b"wh_pfb_create" in _lib.lib.wh_last_error()
    rc = _lib.lib.wh_spectrum_create(ctypes.byref(h), 1)
    assert rc == -1


def test_product_has_no_oracle_or_cpu_fallback():
    """The product package must not import the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "wavecap-sdr_amd", "wavehip")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in txt.replace("no oracle", ""), fn
            assert "ref_np" not in txt, fn
