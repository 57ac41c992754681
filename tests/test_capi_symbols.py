"""CPU: the C-ABI library loads and exports every symbol include/carmel_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "carmel_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(carmel_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from carmel_amd import _capi
    lib = ctypes.CDLL(_capi.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_capi.SYMBOLS) == syms


def test_no_gpu_means_loud_failure():
    """without a device carmel_hip_create must refuse (there is no CPU fallback in the product path)"""
    import numpy as np
    from carmel_amd._capi import lib, ptr
    if lib.carmel_hip_device_count() > 0:
        return
    h = ctypes.c_void_p()
    z = np.zeros(1, np.uint32)
    lw = np.zeros(1)
    rc = lib.carmel_hip_create(ctypes.byref(h), 0, 2, 1, 1, ptr(z), ptr(np.ones(1, np.uint32)), ptr(z), ptr(z), ptr(lw), None)
    assert rc == -2
    assert b"HIP" in lib.carmel_hip_last_error() or b"device" in lib.carmel_hip_last_error()


def test_product_does_not_touch_oracle():
    """the product path must not import, link or call anything under oracle/"""
    for d, _, files in os.walk(os.path.join(ROOT, "carmel_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(d, f)).read()
                for needle in ("import oracle", "from oracle", "oracle/", "liboracle", "orc_", "oracle."):
                    assert needle not in txt, (os.path.join(d, f), needle)
