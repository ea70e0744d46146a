"""The C-ABI library loads and exports every symbol include/watsor_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

from watsor_amd import _lib

HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "watsor_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wz_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libwatsor_hip.so does not export %s" % n


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_no_gpu_calls_fail_cleanly():
    lib = _lib.load()
    assert lib.wz_device_count() >= 0
    h = ctypes.c_void_p()
    rc = lib.wz_create(b"/nonexistent/mi355x.bin", 0, 1, 64, 64, ctypes.byref(h))
    assert rc == _lib.WZ_ENOENT and "not found" in _lib.last_error()
    try:
        _lib.check(rc)
        assert False
    except FileNotFoundError:
        pass
