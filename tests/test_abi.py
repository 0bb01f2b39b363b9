"""The C-ABI library loads on a CPU-only host and exports exactly what include/wavehip.h
declares (no compute calls here)."""

import ctypes
import os
import re

import pytest

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
    # every create() validates its arguments before the first HIP call; a failed create leaves the handle untouched
    import numpy as np
    taps = np.ones(8, dtype=np.float64)
    ftaps = np.ones(63, dtype=np.float32)
    bad = [
        ("wh_chanbank_create", lambda: _lib.lib.wh_chanbank_create(ctypes.byref(h), None)),
        ("wh_lsm_bank_create", lambda: _lib.lib.wh_lsm_bank_create(ctypes.byref(h), 0, 10.0, _lib.dptr(ftaps, "f32"),
                                                                   _lib.dptr(ftaps, "f32"), 1024)),
        ("wh_lsm_bank_create", lambda: _lib.lib.wh_lsm_bank_create(ctypes.byref(h), 4, 1.0, _lib.dptr(ftaps, "f32"),
                                                                   _lib.dptr(ftaps, "f32"), 1024)),
        ("wh_ddc_bank_create", lambda: _lib.lib.wh_ddc_bank_create(ctypes.byref(h), 65, 2400000, _lib.dptr(taps, "f64"),
                                                                   8, 10, None, 0, 1, 4096)),
        ("wh_ddc_create", lambda: _lib.lib.wh_ddc_create(ctypes.byref(h), 0, _lib.dptr(taps, "f64"), 8, 10, None, 0, 1, 4096)),
        ("wh_bch_create", lambda: _lib.lib.wh_bch_create(ctypes.byref(h), None)),
        ("wh_resampler_create", lambda: _lib.lib.wh_resampler_create(ctypes.byref(h), None, 0, 1, 1, 0)),
        ("wh_gardner_bank_create", lambda: _lib.lib.wh_gardner_bank_create(ctypes.byref(h), 0, 10.0, 0.1, 0.01)),
    ]
    for name, call in bad:
        h.value = None
        assert call() == -1, name
        assert name.encode() in _lib.lib.wh_last_error(), (name, _lib.lib.wh_last_error())
        assert not h.value, name
    # run() on a null handle is an argument error too, not a crash
    assert _lib.lib.wh_lsm_bank_run(None, None, 16, 16, None, None, 16, None, None) == -1
    assert _lib.lib.wh_ddc_bank_run(None, None, 16, None, None, None, 16, None) == -1
    assert _lib.lib.wh_bch_decode(None, None, 1, None, None, None, None) == -1


def test_product_has_no_oracle_or_cpu_fallback():
    """The product package must not import the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "wavecap-sdr_amd", "wavehip")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in txt.replace("no oracle", ""), fn
            assert "ref_np" not in txt, fn


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/wavehip.h must be usable from plain C (C11, -pedantic): compile tests/c/abi_smoke.c with gcc, link it
    against libwavehip.so and run it (argument-error path only: no GPU needed)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "wavecap-sdr_amd", "wavehip")
    exe = str(tmp_path / "abi_smoke")
    r = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-o", exe, os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L", libdir, "-lwavehip",
                        f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "wh_pfb_create" in r.stdout, (r.returncode, r.stdout, r.stderr)
