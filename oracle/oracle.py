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
        _declare_more(_lib)
        _declare_harris(_lib)
        _declare_match(_lib)
        _declare_subpix(_lib)
        _declare_cc(_lib)
        _declare_sift(_lib)
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


def _declare_more(L):
    L.vo_or_score_fundamental.restype = None
    L.vo_or_score_fundamental.argtypes = [_f32p, _f32p, ctypes.c_int, _f64p, ctypes.c_int, ctypes.c_double, _i32p, _u8p]
    L.vo_or_score_pnp.restype = None
    L.vo_or_score_pnp.argtypes = [_f32p, _f32p, ctypes.c_int, _f64p, _f64p, ctypes.c_int, ctypes.c_double, _i32p, _u8p]
    L.vo_or_ba_residual.restype = ctypes.c_double
    L.vo_or_ba_residual.argtypes = [_f64p, ctypes.c_int, ctypes.c_int, _i64p, _i64p, _f64p, ctypes.c_int, _f64p, _f64p]
    L.vo_or_ba_jacobian.restype = None
    L.vo_or_ba_jacobian.argtypes = [_f64p, ctypes.c_int, ctypes.c_int, _i64p, _i64p, _f64p, ctypes.c_int, _f64p,
                                    ctypes.c_int, _f64p, _f64p]
    L.vo_or_ba_normal_eq.restype = ctypes.c_double
    L.vo_or_ba_normal_eq.argtypes = [_f64p, ctypes.c_int, ctypes.c_int, _i64p, _i64p, _f64p, ctypes.c_int, _f64p,
                                     ctypes.c_int, _f64p, _f64p, _f64p, _f64p, _f64p]


def _declare_harris(L):
    L.vo_or_harris_response.restype = None
    L.vo_or_harris_response.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, _f32p]
    L.vo_or_nms_topk.restype = ctypes.c_int
    L.vo_or_nms_topk.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _f32p, _f32p]


def _declare_subpix(L):
    L.vo_or_harris_mask.restype = None
    L.vo_or_harris_mask.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_double, _u8p]
    L.vo_or_corner_subpix.restype = None
    L.vo_or_corner_subpix.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, _f32p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_double]


def _declare_cc(L):
    L.vo_or_connected_components.restype = ctypes.c_int
    L.vo_or_connected_components.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, _i32p, _f64p, _i32p,
                                             ctypes.c_int]


SIFT_KP = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                    ("response", np.float32), ("octave", np.int32)])


def _declare_sift(L):
    L.vo_or_sift_detect_compute.restype = ctypes.c_int
    L.vo_or_sift_detect_compute.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_void_p, _f32p, ctypes.c_int]
    L.vo_or_sift_pyramid_image.restype = ctypes.c_int
    L.vo_or_sift_pyramid_image.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, ctypes.POINTER(ctypes.c_int),
                                           ctypes.POINTER(ctypes.c_int)]
    L.vo_or_sift_num_octaves.restype = ctypes.c_int
    L.vo_or_sift_num_octaves.argtypes = [ctypes.c_int, ctypes.c_int]
    L.vo_or_sift_sincos.restype = None
    L.vo_or_sift_sincos.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.vo_or_sift_gauss_kernel.restype = ctypes.c_int
    L.vo_or_sift_gauss_kernel.argtypes = [ctypes.c_double, _f32p]


def _declare_match(L):
    L.vo_or_knn_match2.restype = None
    L.vo_or_knn_match2.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_double, _i32p, _f32p, _u8p]


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


def score_fundamental(m1, m2, F, thr, want_masks=True):
    """-> (counts (H,) i32, masks (H,n) u8 or None)."""
    m1 = np.ascontiguousarray(m1, np.float32).reshape(-1, 2)
    m2 = np.ascontiguousarray(m2, np.float32).reshape(-1, 2)
    F = np.ascontiguousarray(F, np.float64).reshape(-1, 9)
    n, H = m1.shape[0], F.shape[0]
    counts = np.zeros(H, np.int32)
    masks = np.zeros((H, n), np.uint8) if want_masks else None
    lib().vo_or_score_fundamental(_p(m1, _f32p), _p(m2, _f32p), n, _p(F, _f64p), H, float(thr), _p(counts, _i32p),
                                  _p(masks, _u8p) if want_masks else None)
    return counts, masks


def score_pnp(X, x, K, Rt, thr, want_masks=True):
    X = np.ascontiguousarray(X, np.float32).reshape(-1, 3)
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    Rt = np.ascontiguousarray(Rt, np.float64).reshape(-1, 12)
    n, H = X.shape[0], Rt.shape[0]
    counts = np.zeros(H, np.int32)
    masks = np.zeros((H, n), np.uint8) if want_masks else None
    lib().vo_or_score_pnp(_p(X, _f32p), _p(x, _f32p), n, _p(K, _f64p), _p(Rt, _f64p), H, float(thr),
                          _p(counts, _i32p), _p(masks, _u8p) if want_masks else None)
    return counts, masks


def _ba_args(params, F, P, ci, pi, obs, K):
    params = np.ascontiguousarray(params, np.float64)
    ci = np.ascontiguousarray(ci, np.int64); pi = np.ascontiguousarray(pi, np.int64)
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    assert params.size == 6 * F + 3 * P and ci.size == pi.size == obs.shape[0]
    return params, ci, pi, obs, K


def ba_residual(params, F, P, ci, pi, obs, K):
    """-> (r (2O,), cost)."""
    params, ci, pi, obs, K = _ba_args(params, F, P, ci, pi, obs, K)
    O = ci.size
    r = np.zeros(2 * O)
    cost = lib().vo_or_ba_residual(_p(params, _f64p), F, P, _p(ci, _i64p), _p(pi, _i64p), _p(obs, _f64p), O,
                                   _p(K, _f64p), _p(r, _f64p))
    return r, cost


def ba_jacobian(params, F, P, ci, pi, obs, K, obs_start=0):
    """-> (Jc (O,2,6), Jp (O,2,3))."""
    params, ci, pi, obs, K = _ba_args(params, F, P, ci, pi, obs, K)
    O = ci.size
    Jc = np.zeros((O, 2, 6)); Jp = np.zeros((O, 2, 3))
    lib().vo_or_ba_jacobian(_p(params, _f64p), F, P, _p(ci, _i64p), _p(pi, _i64p), _p(obs, _f64p), O, _p(K, _f64p),
                            int(obs_start), _p(Jc, _f64p), _p(Jp, _f64p))
    return Jc, Jp


def ba_normal_eq(params, F, P, ci, pi, obs, K, obs_start=0):
    """-> dict(r, cost, Hcc (F,6,6), Hpp (P,3,3), Hcp (O,6,3), g (6F+3P,))."""
    params, ci, pi, obs, K = _ba_args(params, F, P, ci, pi, obs, K)
    O = ci.size
    r = np.zeros(2 * O); Hcc = np.zeros((F, 6, 6)); Hpp = np.zeros((P, 3, 3)); Hcp = np.zeros((O, 6, 3))
    g = np.zeros(6 * F + 3 * P)
    cost = lib().vo_or_ba_normal_eq(_p(params, _f64p), F, P, _p(ci, _i64p), _p(pi, _i64p), _p(obs, _f64p), O,
                                    _p(K, _f64p), int(obs_start), _p(r, _f64p), _p(Hcc, _f64p), _p(Hpp, _f64p),
                                    _p(Hcp, _f64p), _p(g, _f64p))
    return dict(r=r, cost=cost, Hcc=Hcc, Hpp=Hpp, Hcp=Hcp, g=g)


def harris_response(img, k=0.1, mode=0):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    R = np.zeros((h, w), np.float32)
    lib().vo_or_harris_response(_p(img, _u8p), w, h, w, float(k), int(mode), _p(R, _f32p))
    return R


def nms_topk(R, K, rel_thresh=0.01, radius=1):
    R = np.ascontiguousarray(R, np.float32)
    h, w = R.shape
    xy = np.zeros((K, 2), np.float32); sc = np.zeros(K, np.float32)
    m = lib().vo_or_nms_topk(_p(R, _f32p), w, h, float(rel_thresh), int(radius), int(K), _p(xy, _f32p), _p(sc, _f32p))
    return xy, sc, m


def knn_match2(des1, des2, ratio=0.8):
    """Two nearest train rows per query row: (idx (n1,2) int32, dist (n1,2) float32, keep (n1) uint8)."""
    a = np.ascontiguousarray(des1, np.float32)
    b = np.ascontiguousarray(des2, np.float32)
    n1, n2 = a.shape[0], b.shape[0]
    dim = a.shape[1] if a.ndim == 2 else b.shape[1]
    idx = np.full((n1, 2), -1, np.int32)
    dist = np.full((n1, 2), np.inf, np.float32)
    keep = np.zeros(n1, np.uint8)
    lib().vo_or_knn_match2(_p(a, _f32p), n1, _p(b, _f32p), n2, int(dim), float(ratio), _p(idx, _i32p), _p(dist, _f32p),
                           _p(keep, _u8p))
    return idx, dist, keep


def harris_mask(img, k=0.1, rel_thresh=0.01, mode=0):
    """cornerHarris + dilate + threshold + uint8 of FeatureExtractor.harris -> (H,W) uint8 in {0,255}."""
    R = harris_response(img, k, mode)
    h, w = R.shape
    mask = np.empty((h, w), np.uint8)
    lib().vo_or_harris_mask(_p(R, _f32p), h, w, float(rel_thresh), _p(mask, _u8p))
    return mask


def corner_subpix(img, corners, win=(5, 5), max_iters=50, eps=0.001):
    """cv.cornerSubPix restatement on a uint8 image -> refined (M,2) float32 copy."""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.array(corners, np.float32).reshape(-1, 2).copy()
    h, w = img.shape
    lib().vo_or_corner_subpix(_p(img, _u8p), h, w, w, _p(out, _f32p), out.shape[0], int(win[0]), int(win[1]), int(max_iters),
                              float(eps))
    return out


def connected_components(mask, order=1):
    """8-connected components of a mask (nonzero = foreground) with statistics, as
    cv.connectedComponentsWithStats -> (n_labels, labels (H,W) int32, centroids (n,2) float64 (x, y), areas (n,) int32);
    order 0 numbers components by first pixel, order 1 by first 2x2 block (see vo_oracle_cc.c)."""
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    cap = ((h + 1) // 2) * ((w + 1) // 2) + 1
    labels = np.zeros((h, w), np.int32)
    cen = np.zeros((cap, 2), np.float64)
    areas = np.zeros(cap, np.int32)
    n = lib().vo_or_connected_components(_p(mask, _u8p), h, w, w, int(order), _p(labels, _i32p), _p(cen, _f64p),
                                         _p(areas, _i32p), cap)
    return n, labels, cen[:n].copy(), areas[:n].copy()


def harris_corners(img, k=0.1, rel_thresh=0.01, order=1, win=(5, 5), max_iters=50, eps=0.001):
    """FeatureExtractor.harris end to end (reference vo_pipeline/featureExtraction.py:37-51): mask, components,
    float32 centroids (background first), sub-pixel refinement -> (M,2) float32."""
    mask = harris_mask(img, k, rel_thresh, 0)
    _, _, cen, _ = connected_components(mask, order)
    return corner_subpix(img, np.float32(cen), win, max_iters, eps)


def sift(img, n_octave_layers=4, contrast_threshold=0.03, edge_threshold=10.0, sigma=1.6, max_keypoints=200000,
         want_descriptors=True):
    """cv.SIFT_create(0, layers, contrast, edge, sigma).detectAndCompute(img, None) restated ->
    (keypoints: structured array SIFT_KP (x, y, size, angle, response, octave), descriptors (n,128) float32)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    kp = np.zeros(max_keypoints, SIFT_KP)
    des = np.zeros((max_keypoints, 128), np.float32) if want_descriptors else None
    n = lib().vo_or_sift_detect_compute(_p(img, _u8p), h, w, w, int(n_octave_layers), float(contrast_threshold),
                                        float(edge_threshold), float(sigma), kp.ctypes.data,
                                        _p(des, _f32p) if want_descriptors else None, max_keypoints)
    n = min(n, max_keypoints)
    return kp[:n].copy(), (des[:n].copy() if want_descriptors else None)


def sift_pyramid_image(img, octave, index, dog=False, n_octave_layers=4, sigma=1.6):
    """One image of the SIFT scale space: Gaussian (dog=False, index 0..layers+2) or difference (index 0..layers+1)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    ow, oh = ctypes.c_int(), ctypes.c_int()
    n = lib().vo_or_sift_pyramid_image(_p(img, _u8p), h, w, w, int(n_octave_layers), float(sigma), int(dog), int(octave),
                                       int(index), None, ctypes.byref(ow), ctypes.byref(oh))
    if n < 0:
        raise ValueError("sift_pyramid_image: no such image")
    out = np.zeros((oh.value, ow.value), np.float32)
    lib().vo_or_sift_pyramid_image(_p(img, _u8p), h, w, w, int(n_octave_layers), float(sigma), int(dog), int(octave),
                                   int(index), _p(out, _f32p), ctypes.byref(ow), ctypes.byref(oh))
    return out


def sift_sincos(t):
    """(cos, sin) of float32 radians as the SIFT descriptor's rotation takes them (written-out float64 evaluation, see
    vo_or_sift_sincos); scalar or array -> float32 arrays."""
    t = np.atleast_1d(np.asarray(t, np.float32))
    c, s = np.empty(t.shape, np.float32), np.empty(t.shape, np.float32)
    cc, ss = ctypes.c_float(), ctypes.c_float()
    f = lib().vo_or_sift_sincos
    for i, v in enumerate(t.ravel()):
        f(float(v), ctypes.byref(cc), ctypes.byref(ss))
        c.ravel()[i], s.ravel()[i] = cc.value, ss.value
    return c, s


def sift_gauss_kernel(sigma):
    k = np.zeros(64, np.float32)
    n = lib().vo_or_sift_gauss_kernel(float(sigma), _p(k, _f32p))
    return k[:n].copy()
