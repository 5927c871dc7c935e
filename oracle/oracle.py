"""ctypes front end of oracle/libvo_oracle.so (plain-C restatement of the
reference algorithms).  Test infrastructure: see the header of each
vo_oracle_*.c for the reference lines it follows and its pinning status."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libvo_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.startswith("vo_oracle_") and f.endswith(".c")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _DIR, "-B", "libvo_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _declare(_lib)
    return _lib


_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i16p = ctypes.POINTER(ctypes.c_int16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def _declare(L):
    L.vo_or_pyr_max_level.restype = ctypes.c_int
    L.vo_or_pyr_max_level.argtypes = [ctypes.c_int] * 4
    L.vo_or_pyr_down.restype = None
    L.vo_or_pyr_down.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u8p, ctypes.c_int]
    L.vo_or_scharr.restype = None
    L.vo_or_scharr.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i16p]
    L.vo_or_pyr_level.restype = ctypes.c_int
    L.vo_or_pyr_level.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u8p]
    L.vo_or_klt.restype = ctypes.c_int
    L.vo_or_klt.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, ctypes.c_int,
                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_float,
                            _f32p, _u8p, _f32p]


def _p(a, t):
    return a.ctypes.data_as(t)


def pyr_max_level(w, h, max_level, win=21):
    return lib().vo_or_pyr_max_level(int(w), int(h), int(max_level), int(win))


def pyr_down(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().vo_or_pyr_down(_p(img, _u8p), w, h, w, _p(out, _u8p), out.shape[1])
    return out


def scharr(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().vo_or_scharr(_p(img, _u8p), w, h, w, _p(out, _i16p))
    return out


def pyr_level(img, level):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    lw, lh = w, h
    for _ in range(level):
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
    out = np.empty((lh, lw), np.uint8)
    lib().vo_or_pyr_level(_p(img, _u8p), w, h, w, int(level), _p(out, _u8p))
    return out


def klt(img0, img1, pts, win=21, max_level=8, max_iters=10, eps=0.001, min_eig=1e-2, init_pts=None):
    """calcOpticalFlowPyrLK restatement -> (next_pts (N,2) f32, status (N,1) u8, err (N,1) f32)."""
    img0 = np.ascontiguousarray(img0, dtype=np.uint8)
    img1 = np.ascontiguousarray(img1, dtype=np.uint8)
    assert img0.shape == img1.shape and img0.ndim == 2
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    n = pts.shape[0]
    h, w = img0.shape
    out = np.zeros((n, 2), np.float32)
    st = np.zeros((n, 1), np.uint8)
    err = np.zeros((n, 1), np.float32)
    ip = None
    if init_pts is not None:
        init_pts = np.ascontiguousarray(init_pts, dtype=np.float32).reshape(-1, 2)
        ip = _p(init_pts, _f32p)
    lib().vo_or_klt(_p(img0, _u8p), _p(img1, _u8p), w, h, w, _p(pts, _f32p), ip, n, int(win), int(max_level),
                    int(max_iters), float(eps), float(min_eig), _p(out, _f32p), _p(st, _u8p), _p(err, _f32p))
    return out, st, err
