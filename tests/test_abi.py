"""The C-ABI library loads and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            txt = open(os.path.join(inc, f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", txt))
    return names


def test_header_symbols_exported():
    from visual_odometry_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libvo_hip.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 10
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_python_binding_covers_header():
    from visual_odometry_amd import _lib
    assert set(_lib.SIGNATURES) == _declared()
    _lib.load()


def test_version_and_error_string():
    from visual_odometry_amd import _lib
    lib = _lib.load()
    assert lib.vo_version() >= 100
    assert isinstance(lib.vo_last_error(), bytes)
