"""Object layer over the C-ABI: Context, Pyramid, and the batched operators.

Arrays may be numpy (host memory: the library stages and synchronises) or torch
tensors on the context's GPU (raw device pointers: asynchronous on the stream).
torch is used as a buffer provider only.
"""
import atexit
import ctypes
import sys
import weakref
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import KltParams, VO_MEM_DEVICE, VO_MEM_HOST, check


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return ctypes.c_void_p(x.ptr)
    if _is_torch(x):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(x.ctypes.data)


class DeviceArray:
    """A C-contiguous array in the memory of a context's GPU, owned by the library (vo_dev_alloc): what the pipeline
    keeps on the device between two library calls without bringing in a tensor library.  `[:n]` is a view of the
    leading rows; `numpy()` copies out (synchronises the context's stream)."""

    def __init__(self, ctx: "Context", shape, dtype, _base=None):
        self.ctx, self.shape, self.dtype = ctx, tuple(int(v) for v in shape), np.dtype(dtype)
        self._base = _base
        if _base is None:
            # Buffers go back to a per-context pool of power-of-two sizes when they are closed and are handed out again
            # from there: a 30 MB descriptor matrix per bootstrap image would otherwise cost an allocation and -- worse
            # -- a free that waits for the device, at every keyframe.  Work on the context's stream is ordered, so a
            # reused buffer needs no synchronisation.
            self._room = 1 << max(8, (max(self.nbytes, 1) - 1).bit_length())
            free = ctx._dev_pool.get(self._room)
            if free:
                self.ptr = free.pop()
            else:
                h = ctypes.c_void_p()
                check(ctx.lib.vo_dev_alloc(ctx.handle, self._room, ctypes.byref(h)), "vo_dev_alloc")
                self.ptr = h.value
            _live_objects.add(self)
        else:
            self.ptr = _base.ptr

    @classmethod
    def from_numpy(cls, ctx: "Context", a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(ctx, a.shape, a.dtype)
        check(ctx.lib.vo_dev_upload(ctx.handle, ctypes.c_void_p(d.ptr), ctypes.c_void_p(a.ctypes.data), a.nbytes), "vo_dev_upload")
        return d

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def __len__(self) -> int:
        return self.shape[0]

    def __getitem__(self, key) -> "DeviceArray":
        if not (isinstance(key, slice) and key.start in (None, 0) and key.step in (None, 1)):
            raise IndexError("DeviceArray: only leading-row views a[:n]")
        n = self.shape[0] if key.stop is None else max(0, min(int(key.stop), self.shape[0]))
        return DeviceArray(self.ctx, (n,) + self.shape[1:], self.dtype, _base=self._base if self._base is not None else self)

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            check(self.ctx.lib.vo_dev_download(self.ctx.handle, ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(self.ptr), out.nbytes),
                  "vo_dev_download")
        return out

    def close(self) -> None:
        if self._base is None and self.ptr:
            if self.ctx.handle:
                self.ctx._dev_pool.setdefault(self._room, []).append(self.ptr)
            self.ptr = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One (GPU, HIP stream) pair.  Not re-entrant."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        check(self.lib.vo_ctx_create(int(device), ctypes.c_void_p(stream or 0), ctypes.byref(h)), "vo_ctx_create")
        self.handle = h
        self.device = int(device)
        self._dev_pool = {}          # room in bytes -> device pointers of closed DeviceArrays, ready for reuse
        _live_objects.add(self)

    @classmethod
    def on_own_stream(cls, device: int = 0) -> "Context":
        """Context on a new non-blocking HIP stream that the library owns (vo_ctx_create_own_stream): its work is not
        ordered with the default context's, so the two can run side by side."""
        self = cls.__new__(cls)
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        check(self.lib.vo_ctx_create_own_stream(int(device), ctypes.byref(h)), "vo_ctx_create_own_stream")
        self.handle = h
        self.device = int(device)
        self._dev_pool = {}
        _live_objects.add(self)
        return self

    @classmethod
    def from_torch(cls, device: int = 0) -> "Context":
        """Context on torch's current stream of `device` (so torch ops and events order with it)."""
        import torch
        return cls(device, torch.cuda.current_stream(device).cuda_stream)

    def sync(self) -> None:
        check(self.lib.vo_sync(self.handle), "vo_sync")

    def timer_start(self) -> None:
        check(self.lib.vo_timer_start(self.handle), "vo_timer_start")

    def timer_stop_ms(self) -> float:
        ms = ctypes.c_float()
        check(self.lib.vo_timer_stop_ms(self.handle, ctypes.byref(ms)), "vo_timer_stop_ms")
        return float(ms.value)

    def profile_tracker(self, enable: bool) -> None:
        """While enabled, TrackPnpBatch.begin brackets its tracker launch with a HIP event pair."""
        check(self.lib.vo_profile_tracker(self.handle, int(bool(enable))), "vo_profile_tracker")

    def profile_tracker_ms(self) -> float:
        ms = ctypes.c_float()
        check(self.lib.vo_profile_tracker_ms(self.handle, ctypes.byref(ms)), "vo_profile_tracker_ms")
        return float(ms.value)

    def close(self) -> None:
        if self.handle:
            for ptrs in self._dev_pool.values():
                for ptr in ptrs:
                    self.lib.vo_dev_free(self.handle, ctypes.c_void_p(ptr))
            self._dev_pool.clear()
            self.lib.vo_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # at interpreter shutdown the ordered atexit hook below has already run
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


_live_objects = weakref.WeakSet()
_default_ctx = {}


@atexit.register
def _shutdown():
    """Destroy pyramids and solvers before contexts, while the HIP runtime is still up."""
    objs = list(_live_objects)
    for o in objs:
        if isinstance(o, (Pyramid, BaSolver, DeviceArray)):
            o.close()
    for o in objs:
        if isinstance(o, Context):
            o.close()
    _default_ctx.clear()



def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Pyramid:
    """Gaussian pyramid of `batch` grayscale uint8 images of one size."""

    def __init__(self, ctx: Context, batch: int, height: int, width: int, max_level: int = 8, win: int = 21):
        self.ctx = ctx
        self.batch, self.height, self.width, self.win = int(batch), int(height), int(width), int(win)
        h = ctypes.c_void_p()
        check(ctx.lib.vo_pyr_create(ctx.handle, self.batch, self.height, self.width, int(max_level), self.win,
                                    ctypes.byref(h)), "vo_pyr_create")
        self.handle = h
        self.max_level = ctx.lib.vo_pyr_max_level(h)
        self._keep = None
        _live_objects.add(self)

    def level_shape(self, level: int) -> Tuple[int, int]:
        hh, ww, pp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        check(self.ctx.lib.vo_pyr_level_shape(self.handle, int(level), ctypes.byref(hh), ctypes.byref(ww),
                                              ctypes.byref(pp)), "vo_pyr_level_shape")
        return hh.value, ww.value

    def build(self, imgs) -> "Pyramid":
        """imgs: (H,W) or (B,H,W) uint8, numpy or torch-on-GPU (kept alive by this object)."""
        if _is_torch(imgs):
            import torch
            assert imgs.dtype == torch.uint8 and imgs.is_cuda
            t = imgs if imgs.dim() == 3 else imgs.unsqueeze(0)
            if t.stride(2) != 1:
                t = t.contiguous()
            if tuple(t.shape) != (self.batch, self.height, self.width):
                raise ValueError(f"Pyramid.build: expected {(self.batch, self.height, self.width)}, got {tuple(t.shape)}")
            self._keep = t
            check(self.ctx.lib.vo_pyr_build(self.ctx.handle, self.handle, _ptr(t), t.stride(1),
                                            t.stride(0) if self.batch > 1 else t.stride(1) * self.height,
                                            VO_MEM_DEVICE), "vo_pyr_build")
        else:
            a = np.ascontiguousarray(imgs, dtype=np.uint8)
            if a.ndim == 2:
                a = a[None]
            if a.shape != (self.batch, self.height, self.width):
                raise ValueError(f"Pyramid.build: expected {(self.batch, self.height, self.width)}, got {a.shape}")
            check(self.ctx.lib.vo_pyr_build(self.ctx.handle, self.handle, _ptr(a), a.strides[1], a.strides[0],
                                            VO_MEM_HOST), "vo_pyr_build")
        return self

    def download(self, image: int, level: int) -> np.ndarray:
        h, w = self.level_shape(level)
        out = np.empty((h, w), np.uint8)
        check(self.ctx.lib.vo_pyr_download(self.ctx.handle, self.handle, int(image), int(level), _ptr(out)),
              "vo_pyr_download")
        return out

    def close(self) -> None:
        if self.handle:
            self.ctx.lib.vo_pyr_destroy(self.handle)
            self.handle = None
            self._keep = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def _device_points(t, batch: int, what: str):
    """A (batch, n, 2) float32 contiguous GPU view of keypoints, or ValueError: the library reads raw pointers."""
    import torch
    if not (_is_torch(t) and t.is_cuda and t.dtype == torch.float32):
        raise ValueError(f"{what} must be a float32 GPU tensor")
    if t.numel() % (2 * batch):
        raise ValueError(f"{what}: {tuple(t.shape)} does not reshape to ({batch}, n, 2)")
    p = t.reshape(batch, -1, 2)
    if not p.is_contiguous():
        raise ValueError(f"{what} must be contiguous")
    return p


def klt_track(ctx: Context, prev: Pyramid, nxt: Pyramid, pts, max_iters: int = 10, eps: float = 0.001,
              min_eig: float = 1e-2, init_pts=None, out=None):
    """Batched pyramidal LK.  pts: (B,N,2) or (N,2) float32.

    numpy in -> (next_pts, status, err) numpy, synchronous.
    torch in -> torch outputs on the GPU, asynchronous; `out=(next_pts, status, err)` reuses buffers.
    """
    prm = KltParams(int(max_iters), float(eps), float(min_eig))
    if _is_torch(pts):
        import torch
        p = _device_points(pts, prev.batch, "klt_track: pts")
        n = p.shape[1]
        if init_pts is not None:
            init_pts = _device_points(init_pts, prev.batch, "klt_track: init_pts")
            if init_pts.shape != p.shape:
                raise ValueError(f"klt_track: init_pts {tuple(init_pts.shape)} vs pts {tuple(p.shape)}")
        if out is None:
            out = (torch.empty_like(p), torch.empty((prev.batch, n), dtype=torch.uint8, device=p.device),
                   torch.empty((prev.batch, n), dtype=torch.float32, device=p.device))
        o, s, e = out
        for t, dt, shape, what in ((o, torch.float32, (prev.batch, n, 2), "next_pts"), (s, torch.uint8, (prev.batch, n), "status"),
                                   (e, torch.float32, (prev.batch, n), "err")):
            if not (_is_torch(t) and t.is_cuda and t.dtype == dt and tuple(t.shape) == shape and t.is_contiguous()):
                raise ValueError(f"klt_track: out {what} must be a contiguous {dt} GPU tensor of shape {shape}")
        check(ctx.lib.vo_klt_track(ctx.handle, prev.handle, nxt.handle, _ptr(p), _ptr(init_pts), n,
                                   ctypes.byref(prm), _ptr(o), _ptr(s), _ptr(e), VO_MEM_DEVICE), "vo_klt_track")
        return o, s, e
    p = np.ascontiguousarray(pts, dtype=np.float32).reshape(prev.batch, -1, 2)
    n = p.shape[1]
    ip = None if init_pts is None else np.ascontiguousarray(init_pts, dtype=np.float32).reshape(prev.batch, -1, 2)
    if ip is not None and ip.shape != p.shape:
        raise ValueError(f"klt_track: init_pts {ip.shape} vs pts {p.shape}")
    o = np.zeros_like(p)
    s = np.zeros((prev.batch, n), np.uint8)
    e = np.zeros((prev.batch, n), np.float32)
    check(ctx.lib.vo_klt_track(ctx.handle, prev.handle, nxt.handle, _ptr(p), _ptr(ip), n, ctypes.byref(prm),
                               _ptr(o), _ptr(s), _ptr(e), VO_MEM_HOST), "vo_klt_track")
    return o, s, e


def klt_track_fb(ctx: Context, prev: Pyramid, nxt: Pyramid, pts, lam: float, max_iters: int = 10,
                 eps: float = 0.001, min_eig: float = 1e-2):
    """Forward-backward LK with the bidirectional test; returns (next_pts, mask)."""
    prm = KltParams(int(max_iters), float(eps), float(min_eig))
    if _is_torch(pts):
        import torch
        p = _device_points(pts, prev.batch, "klt_track_fb: pts")
        n = p.shape[1]
        o = torch.empty_like(p)
        m = torch.empty((prev.batch, n), dtype=torch.uint8, device=p.device)
        check(ctx.lib.vo_klt_track_fb(ctx.handle, prev.handle, nxt.handle, _ptr(p), n, ctypes.byref(prm), float(lam),
                                      _ptr(o), _ptr(m), VO_MEM_DEVICE), "vo_klt_track_fb")
        return o, m
    p = np.ascontiguousarray(pts, dtype=np.float32).reshape(prev.batch, -1, 2)
    n = p.shape[1]
    o = np.zeros_like(p)
    m = np.zeros((prev.batch, n), np.uint8)
    check(ctx.lib.vo_klt_track_fb(ctx.handle, prev.handle, nxt.handle, _ptr(p), n, ctypes.byref(prm), float(lam),
                                  _ptr(o), _ptr(m), VO_MEM_HOST), "vo_klt_track_fb")
    return o, m


def _k_host(K) -> np.ndarray:
    if _is_torch(K):
        K = K.detach().cpu().numpy()
    return np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)


def _score(ctx, fn, pa, pb, models, thr, want_masks, K=None, pa_cols=2, ms=9):
    if _is_torch(pa):
        import torch
        pa = pa.reshape(-1, pa_cols).contiguous(); pb = pb.reshape(-1, 2).contiguous()
        models = models.reshape(-1, ms).contiguous()
        if not (pa.dtype == torch.float32 and pb.dtype == torch.float32 and models.dtype == torch.float64):
            raise ValueError(f"{fn}: points must be float32 and models float64")
        n, H = pa.shape[0], models.shape[0]
        counts = torch.empty(H, dtype=torch.int32, device=pa.device)
        masks = torch.empty((H, n), dtype=torch.uint8, device=pa.device) if want_masks else None
        mem = VO_MEM_DEVICE
    else:
        pa = np.ascontiguousarray(pa, np.float32).reshape(-1, pa_cols)
        pb = np.ascontiguousarray(pb, np.float32).reshape(-1, 2)
        models = np.ascontiguousarray(models, np.float64).reshape(-1, ms)
        n, H = pa.shape[0], models.shape[0]
        counts = np.zeros(H, np.int32)
        masks = np.zeros((H, n), np.uint8) if want_masks else None
        mem = VO_MEM_HOST
    if pb.shape[0] != n:
        raise ValueError(f"{fn}: {n} vs {pb.shape[0]} correspondences")
    f = getattr(ctx.lib, fn)
    if K is None:
        rc = f(ctx.handle, _ptr(pa), _ptr(pb), n, _ptr(models), H, float(thr), _ptr(counts), _ptr(masks), mem)
    else:
        Kh = _k_host(K)
        rc = f(ctx.handle, _ptr(pa), _ptr(pb), n, Kh.ctypes.data_as(_lib.c_f64p), _ptr(models), H, float(thr),
               _ptr(counts), _ptr(masks), mem)
    check(rc, fn)
    return counts, masks


def score_fundamental(ctx: Context, m1, m2, F, thr: float, want_masks: bool = True):
    """Inlier counts (H,) int32 and masks (H,n) uint8 of H fundamental-matrix candidates F (H,3,3) f64."""
    return _score(ctx, "vo_score_fundamental", m1, m2, F, thr, want_masks)


def score_pnp(ctx: Context, X, x, K, Rt, thr: float, want_masks: bool = True):
    """Inlier counts and masks of H pose candidates Rt (H,3,4) f64 for 3-D points X (n,3) and pixels x (n,2)."""
    return _score(ctx, "vo_score_pnp", X, x, Rt, thr, want_masks, K=K, pa_cols=3, ms=12)


def _ba_prepare(params, F, P, ci, pi, obs):
    if _is_torch(params):
        import torch
        assert params.dtype == torch.float64 and obs.dtype == torch.float64
        assert ci.dtype == torch.int64 and pi.dtype == torch.int64
        obs = obs.reshape(-1, 2)
        for t in (params, ci, pi, obs):
            assert t.is_cuda and t.is_contiguous()
        return params, ci, pi, obs, VO_MEM_DEVICE
    params = np.ascontiguousarray(params, np.float64).ravel()
    ci = np.ascontiguousarray(ci, np.int64).ravel()
    pi = np.ascontiguousarray(pi, np.int64).ravel()
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
    return params, ci, pi, obs, VO_MEM_HOST


def ba_residual(ctx: Context, params, F: int, P: int, ci, pi, obs, K):
    """Reprojection residual (2O,) f64 and cost = 0.5 r.r  (numpy: float cost; torch: 1-element tensor)."""
    params, ci, pi, obs, mem = _ba_prepare(params, F, P, ci, pi, obs)
    O = int(ci.shape[0])
    if int(params.shape[0]) != 6 * F + 3 * P or int(pi.shape[0]) != O or int(obs.shape[0]) != O:
        raise ValueError("ba_residual: inconsistent sizes")
    Kh = _k_host(K)
    if mem == VO_MEM_DEVICE:
        import torch
        r = torch.empty(2 * O, dtype=torch.float64, device=params.device)
        cost = torch.empty(1, dtype=torch.float64, device=params.device)
    else:
        r = np.zeros(2 * O); cost = np.zeros(1)
    check(ctx.lib.vo_ba_residual(ctx.handle, _ptr(params), F, P, _ptr(ci), _ptr(pi), _ptr(obs), O,
                                 Kh.ctypes.data_as(_lib.c_f64p), _ptr(r), _ptr(cost), mem), "vo_ba_residual")
    return r, (cost if mem == VO_MEM_DEVICE else float(cost[0]))


def ba_normal_eq(ctx: Context, params, F: int, P: int, ci, pi, obs, K, obs_start: int = 0):
    """Residual, cost and the normal equations -> dict(r, cost, Hcc, Hpp, Hcp, g)."""
    params, ci, pi, obs, mem = _ba_prepare(params, F, P, ci, pi, obs)
    O = int(ci.shape[0])
    if int(params.shape[0]) != 6 * F + 3 * P or int(pi.shape[0]) != O or int(obs.shape[0]) != O:
        raise ValueError("ba_normal_eq: inconsistent sizes")
    Kh = _k_host(K)
    if mem == VO_MEM_DEVICE:
        import torch
        mk = lambda *s: torch.empty(s, dtype=torch.float64, device=params.device)
    else:
        mk = lambda *s: np.zeros(s)
    r, cost, Hcc, Hpp, Hcp, g = mk(2 * O), mk(1), mk(F, 6, 6), mk(P, 3, 3), mk(O, 6, 3), mk(6 * F + 3 * P)
    check(ctx.lib.vo_ba_normal_eq(ctx.handle, _ptr(params), F, P, _ptr(ci), _ptr(pi), _ptr(obs), O,
                                  Kh.ctypes.data_as(_lib.c_f64p), int(obs_start), _ptr(r), _ptr(cost), _ptr(Hcc),
                                  _ptr(Hpp), _ptr(Hcp), _ptr(g), mem), "vo_ba_normal_eq")
    return dict(r=r, cost=(cost if mem == VO_MEM_DEVICE else float(cost[0])), Hcc=Hcc, Hpp=Hpp, Hcp=Hcp, g=g)


HARRIS = 0
MIN_EIGEN = 1


def harris_response(ctx: Context, imgs, k: float = 0.1, mode: int = HARRIS):
    """Corner response (B,H,W) float32 of uint8 images (H,W) or (B,H,W)."""
    if _is_torch(imgs):
        import torch
        t = imgs if imgs.dim() == 3 else imgs.unsqueeze(0)
        assert t.dtype == torch.uint8 and t.is_cuda
        if t.stride(2) != 1:
            t = t.contiguous()
        B, H, W = t.shape
        R = torch.empty((B, H, W), dtype=torch.float32, device=t.device)
        check(ctx.lib.vo_harris_response(ctx.handle, _ptr(t), B, H, W, t.stride(1), t.stride(0) if B > 1 else t.stride(1) * H,
                                         2, 3, float(k), int(mode), _ptr(R), VO_MEM_DEVICE), "vo_harris_response")
        return R
    a = np.ascontiguousarray(imgs, dtype=np.uint8)
    if a.ndim == 2:
        a = a[None]
    B, H, W = a.shape
    R = np.zeros((B, H, W), np.float32)
    check(ctx.lib.vo_harris_response(ctx.handle, _ptr(a), B, H, W, a.strides[1], a.strides[0], 2, 3, float(k), int(mode),
                                     _ptr(R), VO_MEM_HOST), "vo_harris_response")
    return R


def nms_topk(ctx: Context, R, K: int, rel_thresh: float = 0.01, radius: int = 1):
    """Strongest K local maxima per image -> (xy (B,K,2) f32, scores (B,K) f32, count (B,) i32)."""
    if _is_torch(R):
        import torch
        t = R if R.dim() == 3 else R.unsqueeze(0)
        assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()
        B, H, W = t.shape
        xy = torch.empty((B, K, 2), dtype=torch.float32, device=t.device)
        sc = torch.empty((B, K), dtype=torch.float32, device=t.device)
        cnt = torch.empty(B, dtype=torch.int32, device=t.device)
        mem = VO_MEM_DEVICE
    else:
        t = np.ascontiguousarray(R, dtype=np.float32)
        if t.ndim == 2:
            t = t[None]
        B, H, W = t.shape
        xy = np.zeros((B, K, 2), np.float32); sc = np.zeros((B, K), np.float32); cnt = np.zeros(B, np.int32)
        mem = VO_MEM_HOST
    check(ctx.lib.vo_nms_topk(ctx.handle, _ptr(t), B, H, W, float(rel_thresh), int(radius), int(K), _ptr(xy), _ptr(sc),
                              _ptr(cnt), mem), "vo_nms_topk")
    return xy, sc, cnt


def detect_corners(ctx: Context, imgs, K: int, rel_thresh: float = 0.01, radius: int = 1, k: float = 0.04,
                   mode: int = MIN_EIGEN):
    """Corner response + non-maximum suppression + top-K in one call over uint8 images (H,W) or
    (B,H,W) -> (xy (B,K,2) f32, scores (B,K) f32, count (B,) i32); same results as harris_response
    followed by nms_topk."""
    if _is_torch(imgs):
        import torch
        t = imgs if imgs.dim() == 3 else imgs.unsqueeze(0)
        assert t.dtype == torch.uint8 and t.is_cuda
        if t.stride(2) != 1:
            t = t.contiguous()
        B, H, W = t.shape
        xy = torch.empty((B, K, 2), dtype=torch.float32, device=t.device)
        sc = torch.empty((B, K), dtype=torch.float32, device=t.device)
        cnt = torch.empty(B, dtype=torch.int32, device=t.device)
        rs, ist, mem = t.stride(1), (t.stride(0) if B > 1 else t.stride(1) * H), VO_MEM_DEVICE
    else:
        t = np.ascontiguousarray(imgs, dtype=np.uint8)
        if t.ndim == 2:
            t = t[None]
        B, H, W = t.shape
        xy = np.zeros((B, K, 2), np.float32); sc = np.zeros((B, K), np.float32); cnt = np.zeros(B, np.int32)
        rs, ist, mem = t.strides[1], t.strides[0], VO_MEM_HOST
    check(ctx.lib.vo_detect_corners(ctx.handle, _ptr(t), B, H, W, rs, ist, float(k), int(mode), float(rel_thresh),
                                    int(radius), int(K), _ptr(xy), _ptr(sc), _ptr(cnt), mem), "vo_detect_corners")
    return xy, sc, cnt


def p3p_solve(ctx: Context, X, x, K, subsets):
    """P3P per 4-point sample -> (Rt (S,3,4) f64, valid (S,) u8).  subsets: (S,4) int32."""
    Kh = _k_host(K)
    if _is_torch(X):
        import torch
        X = X.reshape(-1, 3).contiguous(); x = x.reshape(-1, 2).contiguous()
        subsets = subsets.reshape(-1, 4).to(torch.int32).contiguous()
        S = subsets.shape[0]
        Rt = torch.empty((S, 3, 4), dtype=torch.float64, device=X.device)
        valid = torch.empty(S, dtype=torch.uint8, device=X.device)
        mem = VO_MEM_DEVICE
    else:
        X = np.ascontiguousarray(X, np.float32).reshape(-1, 3); x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
        subsets = np.ascontiguousarray(subsets, np.int32).reshape(-1, 4)
        S = subsets.shape[0]
        Rt = np.zeros((S, 3, 4)); valid = np.zeros(S, np.uint8)
        mem = VO_MEM_HOST
    if x.shape[0] != X.shape[0]:
        raise ValueError("p3p_solve: point sets differ in length")
    check(ctx.lib.vo_p3p_solve(ctx.handle, _ptr(X), _ptr(x), int(X.shape[0]), Kh.ctypes.data_as(_lib.c_f64p), _ptr(subsets),
                               int(S), _ptr(Rt), _ptr(valid), mem), "vo_p3p_solve")
    return Rt, valid


def seven_point_solve(ctx: Context, m1, m2, subsets):
    """7-point algorithm per sample -> (F (S,3,3,3) f64 -- up to three matrices each, nsol (S,) i32)."""
    if _is_torch(m1):
        import torch
        m1 = m1.reshape(-1, 2).contiguous(); m2 = m2.reshape(-1, 2).contiguous()
        subsets = subsets.reshape(-1, 7).to(torch.int32).contiguous()
        S = subsets.shape[0]
        F = torch.empty((S, 3, 3, 3), dtype=torch.float64, device=m1.device)
        ns = torch.empty(S, dtype=torch.int32, device=m1.device)
        mem = VO_MEM_DEVICE
    else:
        m1 = np.ascontiguousarray(m1, np.float32).reshape(-1, 2); m2 = np.ascontiguousarray(m2, np.float32).reshape(-1, 2)
        subsets = np.ascontiguousarray(subsets, np.int32).reshape(-1, 7)
        S = subsets.shape[0]
        F = np.zeros((S, 3, 3, 3)); ns = np.zeros(S, np.int32)
        mem = VO_MEM_HOST
    if m2.shape[0] != m1.shape[0]:
        raise ValueError("seven_point_solve: point sets differ in length")
    check(ctx.lib.vo_seven_point_solve(ctx.handle, _ptr(m1), _ptr(m2), int(m1.shape[0]), _ptr(subsets), int(S), _ptr(F),
                                       _ptr(ns), mem), "vo_seven_point_solve")
    return F, ns


def triangulate(ctx: Context, pts1, pts2, M1, M2):
    """DLT triangulation of homogeneous pixel pairs (n,3) float32 -> (n,4) float32 (last column 1).
    [p]x M is formed in float32 for a float32 projection matrix, like the reference's numpy product."""
    f1 = getattr(M1, "dtype", None) == np.float32
    f2 = getattr(M2, "dtype", None) == np.float32
    A = np.ascontiguousarray(M1, np.float64).reshape(3, 4)
    B = np.ascontiguousarray(M2, np.float64).reshape(3, 4)
    if _is_torch(pts1):
        import torch
        pts1 = pts1.reshape(-1, 3).contiguous(); pts2 = pts2.reshape(-1, 3).contiguous()
        out = torch.empty((pts1.shape[0], 4), dtype=torch.float32, device=pts1.device)
        mem = VO_MEM_DEVICE
    else:
        pts1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 3)
        pts2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 3)
        out = np.zeros((pts1.shape[0], 4), np.float32)
        mem = VO_MEM_HOST
    if pts2.shape[0] != pts1.shape[0]:
        raise ValueError("triangulate: point sets differ in length")
    check(ctx.lib.vo_triangulate(ctx.handle, _ptr(pts1), _ptr(pts2), int(pts1.shape[0]), A.ctypes.data_as(_lib.c_f64p),
                                 B.ctypes.data_as(_lib.c_f64p), int(f1), int(f2), _ptr(out), mem), "vo_triangulate")
    return out


def cheirality_vote(ctx: Context, pts1, pts2, M1, M2, T_hat):
    """Triangulate every pair under each of H candidate motions and count the points in front of both
    cameras, in one launch.  M1 (3,4); M2 (H,3,4) projection matrices; T_hat (H,3,4) motions [R|t].
    -> (clouds (H,n,4) float32, in_front (H,) int32); numpy in -> numpy out with one synchronisation."""
    f1 = getattr(M1, "dtype", None) == np.float32
    f2 = getattr(M2, "dtype", None) == np.float32
    A = np.ascontiguousarray(M1, np.float64).reshape(3, 4)
    B = np.ascontiguousarray(M2, np.float64).reshape(-1, 3, 4)
    T = np.ascontiguousarray(T_hat, np.float64).reshape(-1, 3, 4)
    H = B.shape[0]
    if T.shape[0] != H:
        raise ValueError("cheirality_vote: M2 and T_hat differ in length")
    if _is_torch(pts1):
        import torch
        pts1 = pts1.reshape(-1, 3).contiguous(); pts2 = pts2.reshape(-1, 3).contiguous()
        if pts1.dtype != torch.float32 or pts2.dtype != torch.float32:
            raise ValueError("cheirality_vote: device points must be float32")
        clouds = torch.empty((H, pts1.shape[0], 4), dtype=torch.float32, device=pts1.device)
        votes = torch.empty(H, dtype=torch.int32, device=pts1.device)
        mem = VO_MEM_DEVICE
    else:
        pts1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 3)
        pts2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 3)
        clouds = np.zeros((H, pts1.shape[0], 4), np.float32)
        votes = np.zeros(H, np.int32)
        mem = VO_MEM_HOST
    if pts2.shape[0] != pts1.shape[0]:
        raise ValueError("cheirality_vote: point sets differ in length")
    check(ctx.lib.vo_cheirality_vote(ctx.handle, _ptr(pts1), _ptr(pts2), int(pts1.shape[0]), A.ctypes.data_as(_lib.c_f64p),
                                     B.ctypes.data_as(_lib.c_f64p), T.ctypes.data_as(_lib.c_f64p), int(H), int(f1), int(f2),
                                     _ptr(clouds), _ptr(votes), mem), "vo_cheirality_vote")
    return clouds, votes


def harris_mask(ctx: Context, imgs, k: float = 0.1, rel_thresh: float = 0.01, mode: int = HARRIS):
    """255 where the 3x3 maximum of the corner response exceeds rel_thresh * its image maximum, else 0:
    cornerHarris + dilate + threshold + uint8 of the reference's FeatureExtractor.harris.  imgs: (B,H,W) or
    (H,W) uint8, numpy or torch -> mask of the same shape and kind."""
    single = imgs.ndim == 2
    if _is_torch(imgs):
        import torch
        a = imgs if not single else imgs[None]
        if a.dtype != torch.uint8 or a.stride(-1) != 1:
            raise ValueError("harris_mask: device images must be uint8 with unit pixel stride")
        B, H, W = a.shape
        rs, ist = a.stride(1), (a.stride(0) if B > 1 else a.stride(1) * H)
        mask = torch.empty((B, H, W), dtype=torch.uint8, device=a.device)
        mem = VO_MEM_DEVICE
    else:
        a = np.ascontiguousarray(imgs, np.uint8)
        a = a if not single else a[None]
        B, H, W = a.shape
        rs, ist = W, W * H
        mask = np.empty((B, H, W), np.uint8)
        mem = VO_MEM_HOST
    check(ctx.lib.vo_harris_mask(ctx.handle, _ptr(a), int(B), int(H), int(W), int(rs), int(ist), float(k), int(mode),
                                 float(rel_thresh), _ptr(mask), mem), "vo_harris_mask")
    return mask[0] if single else mask


def corner_subpix(ctx: Context, img, corners, win=(5, 5), max_iters: int = 50, eps: float = 0.001):
    """cv.cornerSubPix on one grayscale uint8 image: corners (M,2) float32 (x, y) -> refined copy."""
    if _is_torch(img):
        import torch
        if not _is_torch(corners) or img.dtype != torch.uint8 or img.stride(-1) != 1 or corners.dtype != torch.float32:
            raise ValueError("corner_subpix: device arguments must be a uint8 image and float32 corners")
        H, W = img.shape
        rs = img.stride(0)
        out = corners.reshape(-1, 2).contiguous().clone()
        mem = VO_MEM_DEVICE
    else:
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        rs = W
        out = np.array(corners, np.float32).reshape(-1, 2).copy()
        mem = VO_MEM_HOST
    check(ctx.lib.vo_corner_subpix(ctx.handle, _ptr(img), int(H), int(W), int(rs), _ptr(out), int(out.shape[0]), int(win[0]),
                                   int(win[1]), int(max_iters), float(eps), mem), "vo_corner_subpix")
    return out


CC_FIRST_PIXEL = 0
CC_FIRST_BLOCK = 1


def connected_components(ctx: Context, mask, order: int = CC_FIRST_BLOCK, want_labels: bool = True):
    """8-connected components of one uint8 mask (nonzero = foreground) with statistics, as
    cv.connectedComponentsWithStats -> (n_labels, labels (H,W) int32 or None, centroids (n,2) float64 (x, y),
    areas (n,) int32); label 0 is the background.  order: CC_FIRST_PIXEL or CC_FIRST_BLOCK numbering."""
    n = ctypes.c_int32()
    if _is_torch(mask):
        import torch
        if mask.dtype != torch.uint8 or mask.dim() != 2 or mask.stride(1) != 1:
            raise ValueError("connected_components: device mask must be a 2-D uint8 tensor with unit pixel stride")
        H, W = mask.shape
        cap = ((H + 1) // 2) * ((W + 1) // 2) + 1
        labels = torch.empty((H, W), dtype=torch.int32, device=mask.device) if want_labels else None
        cen = torch.empty((cap, 2), dtype=torch.float64, device=mask.device)
        areas = torch.empty(cap, dtype=torch.int32, device=mask.device)
        rs, mem = mask.stride(0), VO_MEM_DEVICE
    else:
        mask = np.ascontiguousarray(mask, np.uint8)
        if mask.ndim != 2:
            raise ValueError("connected_components: mask must be 2-D")
        H, W = mask.shape
        cap = ((H + 1) // 2) * ((W + 1) // 2) + 1
        labels = np.empty((H, W), np.int32) if want_labels else None
        cen = np.empty((cap, 2), np.float64)
        areas = np.empty(cap, np.int32)
        rs, mem = W, VO_MEM_HOST
    check(ctx.lib.vo_connected_components(ctx.handle, _ptr(mask), int(H), int(W), int(rs), int(order), _ptr(labels), _ptr(cen),
                                          _ptr(areas), int(cap), ctypes.byref(n), mem), "vo_connected_components")
    return int(n.value), labels, cen[:n.value], areas[:n.value]


def harris_corners(ctx: Context, img, k: float = 0.1, rel_thresh: float = 0.01, order: int = CC_FIRST_BLOCK, win=(5, 5),
                   max_iters: int = 50, eps: float = 0.001, max_corners: int = 0):
    """FeatureExtractor.harris of the reference in one call: Harris mask, connected components, float32 centroids
    (background first), cornerSubPix -> (M,2) float32 (x, y).  max_corners = 0: room for every possible label."""
    n = ctypes.c_int32()
    if _is_torch(img):
        import torch
        if img.dtype != torch.uint8 or img.dim() != 2 or img.stride(1) != 1:
            raise ValueError("harris_corners: device image must be a 2-D uint8 tensor with unit pixel stride")
        H, W = img.shape
        cap = int(max_corners) or ((H + 1) // 2) * ((W + 1) // 2) + 1
        out = torch.empty((cap, 2), dtype=torch.float32, device=img.device)
        rs, mem = img.stride(0), VO_MEM_DEVICE
    else:
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        cap = int(max_corners) or ((H + 1) // 2) * ((W + 1) // 2) + 1
        out = np.empty((cap, 2), np.float32)
        rs, mem = W, VO_MEM_HOST
    check(ctx.lib.vo_harris_corners(ctx.handle, _ptr(img), int(H), int(W), int(rs), float(k), float(rel_thresh), int(order),
                                    int(win[0]), int(win[1]), int(max_iters), float(eps), _ptr(out), int(cap), ctypes.byref(n),
                                    mem), "vo_harris_corners")
    return out[:min(int(n.value), cap)]


SIFT_KEYPOINT = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                          ("response", np.float32), ("octave", np.int32)])


SIFT_FIRST_ROOM = 65536          # keypoints a first sift_detect_compute attempt has room for


def sift_detect_compute(ctx: Context, img, n_octave_layers: int = 4, contrast_threshold: float = 0.03, edge_threshold: float = 10.0,
                        sigma: float = 1.6, max_keypoints: Optional[int] = None, want_descriptors: bool = True,
                        device_descriptors: bool = False):
    """SIFT keypoints and descriptors of one grayscale uint8 image, as cv.SIFT_create(0, layers, contrast, edge,
    sigma).detectAndCompute(img, None) -> (keypoints: structured numpy array SIFT_KEYPOINT, descriptors (n,128) float32
    -- numpy for a numpy image, a GPU tensor for a GPU image, a DeviceArray for a numpy image with device_descriptors
    -- or None).  max_keypoints = None returns all keypoints, as cv2 does (an image with more than the room of the
    first attempt is described again with the room it asked for); a number keeps the first that many in cv2's order
    (x-major: the left of the image)."""
    n = ctypes.c_int32()
    truncate = max_keypoints is not None
    if max_keypoints is None:
        max_keypoints = SIFT_FIRST_ROOM
    while True:
        kp = np.empty(int(max_keypoints), SIFT_KEYPOINT)
        if _is_torch(img):
            import torch
            if img.dtype != torch.uint8 or img.dim() != 2 or img.stride(1) != 1:
                raise ValueError("sift_detect_compute: device image must be a 2-D uint8 tensor with unit pixel stride")
            H, W = img.shape
            des = torch.empty((int(max_keypoints), 128), dtype=torch.float32, device=img.device) if want_descriptors else None
            src, rs, mem = img, img.stride(0), VO_MEM_DEVICE
        else:
            img = np.ascontiguousarray(img, np.uint8)
            if img.ndim != 2:
                raise ValueError("sift_detect_compute: image must be 2-D")
            H, W = img.shape
            if device_descriptors and want_descriptors:
                src = DeviceArray.from_numpy(ctx, img)
                des = DeviceArray(ctx, (int(max_keypoints), 128), np.float32)
                rs, mem = W, VO_MEM_DEVICE
            else:
                des = np.empty((int(max_keypoints), 128), np.float32) if want_descriptors else None
                src, rs, mem = img, W, VO_MEM_HOST
        check(ctx.lib.vo_sift_detect_compute(ctx.handle, _ptr(src), int(H), int(W), int(rs), int(n_octave_layers),
                                             float(contrast_threshold), float(edge_threshold), float(sigma), _ptr(kp), _ptr(des),
                                             int(max_keypoints), ctypes.byref(n), mem), "vo_sift_detect_compute")
        if truncate or int(n.value) <= int(max_keypoints):
            break
        max_keypoints = int(n.value)          # more keypoints than room: again, with room for all of them
    m = min(int(n.value), int(max_keypoints))
    return kp[:m], (des[:m] if want_descriptors else None)


def sift_scale_space_image(ctx: Context, img, octave: int, index: int, dog: bool = False, n_octave_layers: int = 4,
                           sigma: float = 1.6) -> np.ndarray:
    """One image of the SIFT scale space (tests / debugging): Gaussian, or difference of Gaussians with dog=True."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    ow, oh = ctypes.c_int(), ctypes.c_int()
    args = (ctx.handle, _ptr(img), int(H), int(W), int(W), int(n_octave_layers), float(sigma), int(dog), int(octave), int(index))
    check(ctx.lib.vo_sift_scale_space_image(*args, None, ctypes.byref(ow), ctypes.byref(oh), VO_MEM_HOST), "vo_sift_scale_space_image")
    out = np.empty((oh.value, ow.value), np.float32)
    check(ctx.lib.vo_sift_scale_space_image(*args, _ptr(out), ctypes.byref(ow), ctypes.byref(oh), VO_MEM_HOST), "vo_sift_scale_space_image")
    return out


def knn_match2(ctx: Context, des1, des2, ratio: float = 0.8):
    """Two nearest rows of des2 (n2,128) for every row of des1 (n1,128), exhaustive float32 L2 search.
    Returns (idx (n1,2) int32, dist (n1,2) float32, keep (n1) uint8); keep = dist[:,0] < ratio * dist[:,1].
    A missing neighbour is index -1 at distance inf."""
    if isinstance(des1, DeviceArray) or isinstance(des2, DeviceArray):
        # library-owned device descriptors: the three small results come back as numpy arrays
        if not (isinstance(des1, DeviceArray) and isinstance(des2, DeviceArray)) or des1.dtype != np.float32 or des2.dtype != np.float32:
            raise ValueError("knn_match2: device descriptors must both be float32 DeviceArrays")
        if des1.ndim != 2 or des2.ndim != 2 or des1.shape[1] != des2.shape[1]:
            raise ValueError("knn_match2: descriptors must be (n1,d) and (n2,d), got %s and %s" % (des1.shape, des2.shape))
        n1, n2, dim = des1.shape[0], des2.shape[0], des1.shape[1]
        # one buffer: idx (n1,2) int32 | dist (n1,2) float32 | keep (n1) uint8
        buf = DeviceArray(ctx, (n1 * 17 + 16,), np.uint8)
        check(ctx.lib.vo_knn_match2(ctx.handle, _ptr(des1), n1, _ptr(des2), n2, dim, float(ratio), ctypes.c_void_p(buf.ptr),
                                    ctypes.c_void_p(buf.ptr + n1 * 8), ctypes.c_void_p(buf.ptr + n1 * 16), VO_MEM_DEVICE), "vo_knn_match2")
        raw = buf.numpy()
        buf.close()
        return (raw[:n1 * 8].view(np.int32).reshape(n1, 2), raw[n1 * 8:n1 * 16].view(np.float32).reshape(n1, 2),
                raw[n1 * 16:n1 * 17].copy())
    if _is_torch(des1):
        import torch
        if not _is_torch(des2) or des1.dtype != torch.float32 or des2.dtype != torch.float32:
            raise ValueError("knn_match2: device descriptors must both be float32 tensors")
        des1, des2 = des1.contiguous(), des2.contiguous()
        mem = VO_MEM_DEVICE
    else:
        des1 = np.ascontiguousarray(des1, np.float32)
        des2 = np.ascontiguousarray(des2, np.float32)
        mem = VO_MEM_HOST
    if des1.ndim != 2 or des2.ndim != 2 or des1.shape[1] != des2.shape[1]:
        raise ValueError("knn_match2: descriptors must be (n1,d) and (n2,d), got %s and %s" % (tuple(des1.shape), tuple(des2.shape)))
    n1, n2, dim = int(des1.shape[0]), int(des2.shape[0]), int(des1.shape[1])
    if mem == VO_MEM_DEVICE:
        idx = torch.empty((n1, 2), dtype=torch.int32, device=des1.device)
        dist = torch.empty((n1, 2), dtype=torch.float32, device=des1.device)
        keep = torch.empty((n1,), dtype=torch.uint8, device=des1.device)
    else:
        idx = np.empty((n1, 2), np.int32)
        dist = np.empty((n1, 2), np.float32)
        keep = np.empty((n1,), np.uint8)
    check(ctx.lib.vo_knn_match2(ctx.handle, _ptr(des1), n1, _ptr(des2), n2, dim, float(ratio), _ptr(idx), _ptr(dist),
                                _ptr(keep), mem), "vo_knn_match2")
    return idx, dist, keep


class BaSolver:
    """Device-resident Levenberg-Marquardt state of one bundle-adjustment window (vo_ba_*).
    ci, pi, obs may be numpy arrays or torch tensors on the context's device; K is host."""

    def __init__(self, ctx: Context, F: int, P: int, ci, pi, obs, K, obs_start: int = 0):
        self.ctx, self.F, self.P = ctx, int(F), int(P)
        self.handle = None
        Kd = np.ascontiguousarray(K, np.float64).reshape(9)
        if _is_torch(ci):
            import torch
            if not (_is_torch(pi) and _is_torch(obs)):
                raise ValueError("BaSolver: ci, pi, obs must live in the same memory")
            if ci.dtype != torch.int64 or pi.dtype != torch.int64 or obs.dtype != torch.float64:
                raise ValueError("BaSolver: device inputs must be int64 indices and float64 observations")
            ci, pi, obs = ci.contiguous(), pi.contiguous(), obs.contiguous()
            O, mem = int(ci.numel()), VO_MEM_DEVICE
        else:
            ci = np.ascontiguousarray(ci, np.int64).ravel()
            pi = np.ascontiguousarray(pi, np.int64).ravel()
            obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
            O, mem = int(ci.size), VO_MEM_HOST
        if int(pi.shape[0] if not _is_torch(pi) else pi.numel()) != O or int(obs.shape[0]) != O:
            raise ValueError("BaSolver: ci, pi and obs differ in length")
        self.O = O
        h = ctypes.c_void_p()
        check(ctx.lib.vo_ba_create(ctx.handle, self.F, self.P, O, _ptr(ci), _ptr(pi), _ptr(obs), Kd.ctypes.data_as(_lib.c_f64p),
                                   int(obs_start), mem, ctypes.byref(h)), "vo_ba_create")
        self.handle = h
        _live_objects.add(self)

    @classmethod
    def filtered(cls, ctx: Context, F: int, P: int, ci, pi, obs, K, x0, max_sq_error: float, n_frozen_cams: int) -> "BaSolver":
        """Solver of the observations that survive the outlier pre-filter at x0 (vo_ba_create_filtered): residual,
        filter, compaction and landmark lists on the device, one upload of the observations.  ci, pi, obs, x0: numpy arrays or
        torch tensors on the context's device (int64, int64, float64 or float32 (O,2), float64 6F+3P).  The result's .O and .obs_start describe the kept observations."""
        self = cls.__new__(cls)
        self.ctx, self.F, self.P = ctx, int(F), int(P)
        self.handle = None
        Kd = np.ascontiguousarray(K, np.float64).reshape(9)
        if _is_torch(ci):
            import torch
            if not (_is_torch(pi) and _is_torch(obs) and _is_torch(x0)):
                raise ValueError("BaSolver.filtered: ci, pi, obs and x0 must live in the same memory")
            if ci.dtype != torch.int64 or pi.dtype != torch.int64 or obs.dtype not in (torch.float32, torch.float64) or \
                    x0.dtype != torch.float64:
                raise ValueError("BaSolver.filtered: device inputs must be int64 indices, float32/float64 observations, float64 x0")
            ci, pi, obs, x0 = ci.contiguous(), pi.contiguous(), obs.contiguous(), x0.contiguous()
            f32 = obs.dtype == torch.float32
            O, n_obs, n_pi, n_x, mem = int(ci.numel()), int(obs.numel()) // 2, int(pi.numel()), int(x0.numel()), VO_MEM_DEVICE
        else:
            ci = np.ascontiguousarray(ci, np.int64).ravel()
            pi = np.ascontiguousarray(pi, np.int64).ravel()
            obs = np.asarray(obs)
            f32 = obs.dtype == np.float32               # the pipeline's keypoints: sent as they are, widened on the device
            obs = np.ascontiguousarray(obs, np.float32 if f32 else np.float64).reshape(-1, 2)
            x0 = np.ascontiguousarray(x0, np.float64).ravel()
            O, n_obs, n_pi, n_x, mem = int(ci.size), int(obs.shape[0]), int(pi.size), int(x0.size), VO_MEM_HOST
        if n_pi != O or n_obs != O:
            raise ValueError("BaSolver.filtered: ci, pi and obs differ in length")
        if n_x != 6 * self.F + 3 * self.P:
            raise ValueError(f"BaSolver.filtered: x0 has {n_x} entries, expected {6 * self.F + 3 * self.P}")
        h = ctypes.c_void_p()
        kept, start = ctypes.c_int32(), ctypes.c_int32()
        check(ctx.lib.vo_ba_create_filtered(ctx.handle, self.F, self.P, O, _ptr(ci), _ptr(pi), _ptr(obs), int(f32),
                                            Kd.ctypes.data_as(_lib.c_f64p), _ptr(x0), float(max_sq_error), int(n_frozen_cams),
                                            mem, ctypes.byref(h), ctypes.byref(kept), ctypes.byref(start)),
              "vo_ba_create_filtered")
        self.handle = h
        self.O, self.obs_start = int(kept.value), int(start.value)
        _live_objects.add(self)
        return self

    def linearize(self, x) -> float:
        """Set the current parameter vector (6F+3P float64) and build residual + normal equations; returns the cost."""
        cost = ctypes.c_double()
        if _is_torch(x):
            x = x.contiguous()
            mem = VO_MEM_DEVICE
            n = int(x.numel())
        else:
            x = np.ascontiguousarray(x, np.float64).ravel()
            mem = VO_MEM_HOST
            n = int(x.size)
        if n != 6 * self.F + 3 * self.P:
            raise ValueError(f"BaSolver.linearize: parameter vector has {n} entries, expected {6 * self.F + 3 * self.P}")
        check(self.ctx.lib.vo_ba_linearize(self.ctx.handle, self.handle, _ptr(x), mem, ctypes.byref(cost)), "vo_ba_linearize")
        return cost.value

    def try_step(self, lam: float) -> Tuple[float, bool]:
        """Candidate = current + LM step for damping lam -> (cost at the candidate, reduced system solved)."""
        cost, ok = ctypes.c_double(), ctypes.c_int()
        check(self.ctx.lib.vo_ba_try_step(self.ctx.handle, self.handle, float(lam), ctypes.byref(cost), ctypes.byref(ok)),
              "vo_ba_try_step")
        return cost.value, bool(ok.value)

    def accept(self) -> None:
        check(self.ctx.lib.vo_ba_accept(self.ctx.handle, self.handle), "vo_ba_accept")

    def solve(self, x0, max_iters: int = 50, ftol: float = 1e-4, lam0: float = 1e-3, stop_on_flat_reject: bool = True):
        """The whole Levenberg-Marquardt loop from x0 with the accept / reject decisions on the device (vo_ba_solve):
        trial steps are queued several at a time, the host reads the loop's record once per round -> (x (6F+3P) float64,
        stats: cost_start, cost_final, lam, trial_steps, accepted, iterations, stop_reason)."""
        x0 = np.ascontiguousarray(x0, np.float64).ravel()
        if x0.size != 6 * self.F + 3 * self.P:
            raise ValueError(f"BaSolver.solve: parameter vector has {x0.size} entries, expected {6 * self.F + 3 * self.P}")
        x = np.empty_like(x0)
        st = _lib.BaLmStats()
        check(self.ctx.lib.vo_ba_solve(self.ctx.handle, self.handle, _ptr(x0), int(max_iters), float(ftol), float(lam0),
                                       int(bool(stop_on_flat_reject)), _ptr(x), VO_MEM_HOST, ctypes.byref(st)), "vo_ba_solve")
        return x, {k: getattr(st, k) for k, _ in st._fields_}

    def solve_begin(self, x0, max_iters: int = 50, ftol: float = 1e-4, lam0: float = 1e-3, stop_on_flat_reject: bool = True,
                    first_round: int = 0) -> None:
        """First half of solve (vo_ba_solve_begin): x0 goes up, the first round of trial steps is queued (first_round of
        them; 0: the five that solve queues), nothing is waited for.  solve_finish returns the result; other contexts
        may be used in between."""
        x0 = np.ascontiguousarray(x0, np.float64).ravel()
        if x0.size != 6 * self.F + 3 * self.P:
            raise ValueError(f"BaSolver.solve_begin: parameter vector has {x0.size} entries, expected {6 * self.F + 3 * self.P}")
        check(self.ctx.lib.vo_ba_solve_begin(self.ctx.handle, self.handle, _ptr(x0), int(max_iters), float(ftol), float(lam0),
                                             int(bool(stop_on_flat_reject)), int(first_round), VO_MEM_HOST), "vo_ba_solve_begin")

    def solve_finish(self, workspace: Optional["HostWorkspace"] = None):
        """Second half of solve (vo_ba_solve_finish) -> (x (6F+3P) float64, stats).  workspace: x is a view of its
        buffer, valid until the next call with it."""
        n = 6 * self.F + 3 * self.P
        x = np.empty(n, np.float64) if workspace is None else workspace.take("solve x", n, np.float64)
        st = _lib.BaLmStats()
        check(self.ctx.lib.vo_ba_solve_finish(self.ctx.handle, self.handle, _ptr(x), VO_MEM_HOST, ctypes.byref(st)), "vo_ba_solve_finish")
        return x, {k: getattr(st, k) for k, _ in st._fields_}

    def params(self, candidate: bool = False) -> np.ndarray:
        x = np.empty(6 * self.F + 3 * self.P, np.float64)
        check(self.ctx.lib.vo_ba_params(self.ctx.handle, self.handle, int(candidate), _ptr(x), VO_MEM_HOST), "vo_ba_params")
        return x

    def close(self) -> None:
        if self.handle:
            self.ctx.lib.vo_ba_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def ba_camera_params(M) -> np.ndarray:
    """Host-side (vo_ba_camera_params): window poses (F, 3, 4) -> the solver's camera parameters (F, 6) float64,
    rotation vector and translation per frame (cv.Rodrigues of every rotation block)."""
    M = np.ascontiguousarray(M, np.float64)
    if M.ndim != 3 or M.shape[1:] != (3, 4):
        raise ValueError("ba_camera_params: poses must be (F, 3, 4), got %s" % (M.shape,))
    out = np.empty((M.shape[0], 6), np.float64)
    check(_lib.load().vo_ba_camera_params(M.shape[0], _ptr(M), _ptr(out)), "vo_ba_camera_params")
    return out


class HostWorkspace:
    """Host arrays that a caller fills again and again with results of growing size (a bundle-adjustment window per
    keyframe): `take(name, n, dtype, width)` hands out the first n rows of a buffer kept under `name`, which grows
    by doubling.  A fresh multi-megabyte numpy array is fresh pages from the kernel, every one a fault at its first
    write -- more time than the pass that fills them; what take() returns is valid until the next take() of that name."""

    def __init__(self):
        self._buffers = {}

    def take(self, name: str, n: int, dtype, width: int = 0) -> np.ndarray:
        dtype = np.dtype(dtype)
        per_row = max(1, width)
        buf = self._buffers.get(name)
        if buf is None or buf.dtype != dtype or buf.size < n * per_row:
            room = max(1 << 14, 2 * n * per_row)
            buf = self._buffers[name] = np.empty(room, dtype)
        out = buf[:n * per_row]
        return out.reshape(n, width) if width else out


def ba_window_assemble(kps_per_frame, ids_per_frame, workspace: Optional[HostWorkspace] = None):
    """Host-side (vo_ba_window_assemble): the observation arrays of a bundle-adjustment window from per-frame lists of
    keypoints ((n_f, 2) float32) and trajectory ids ((n_f,) int32) -> (camera_indices (O,) int64 ascending,
    point_indices (O,) int64, keypoints (O, 2) float32, trajectory id of every window landmark in order of first
    appearance (P,) int32), or None when the ids are too sparse for the library's table (the caller then numbers
    them by sorting).  workspace: the four arrays are views of its buffers, valid until the next call with it."""
    lib = _lib.load()
    F = len(ids_per_frame)
    if len(kps_per_frame) != F:
        raise ValueError("ba_window_assemble: keypoint and id lists differ in length")
    # a window has a hundred frames and this runs at every keyframe: arrays that already are what the library reads
    # (the track store's are) are passed as they are, the others converted
    f32, i32 = np.dtype(np.float32), np.dtype(np.int32)
    kps, ids = list(kps_per_frame), list(ids_per_frame)
    counts = np.empty(F, np.int32)
    for f in range(F):
        k, t = kps[f], ids[f]
        if not (type(k) is np.ndarray and k.dtype is f32 and k.ndim == 2 and k.flags.c_contiguous):
            k = kps[f] = np.ascontiguousarray(k, np.float32).reshape(-1, 2)
        if not (type(t) is np.ndarray and t.dtype is i32 and t.ndim == 1 and t.flags.c_contiguous):
            t = ids[f] = np.ascontiguousarray(t, np.int32).ravel()
        if k.shape[0] != t.shape[0] or k.shape[1] != 2:
            raise ValueError("ba_window_assemble: a frame's keypoints and ids differ in length")
        counts[f] = t.shape[0]
    O = int(counts.sum())
    kp_ptrs = (ctypes.c_void_p * max(F, 1))(*[k.ctypes.data for k in kps])
    id_ptrs = (ctypes.c_void_p * max(F, 1))(*[t.ctypes.data for t in ids])
    if workspace is None:
        ci, pi = np.empty(O, np.int64), np.empty(O, np.int64)
        obs, traj = np.empty((O, 2), np.float32), np.empty(O, np.int32)
    else:
        ci, pi = workspace.take("window ci", O, np.int64), workspace.take("window pi", O, np.int64)
        obs, traj = workspace.take("window obs", O, np.float32, 2), workspace.take("window traj", O, np.int32)
    n_points = ctypes.c_int32()
    rc = lib.vo_ba_window_assemble(F, _ptr(counts), ctypes.cast(kp_ptrs, ctypes.c_void_p), ctypes.cast(id_ptrs, ctypes.c_void_p),
                                   _ptr(ci), _ptr(pi), _ptr(obs), _ptr(traj), ctypes.byref(n_points))
    if rc == 1:
        return None
    check(rc, "vo_ba_window_assemble")
    return ci, pi, obs, traj[:n_points.value]


def ransac_draw_subsets(rng_state: int, count: int, model_points: int, n_subsets: int, m1=None, m2=None,
                        max_attempts: int = 10000) -> Tuple[np.ndarray, int]:
    """Host-side: n_subsets index sets drawn as OpenCV's RANSAC loop draws them -> (subsets (n_subsets, model_points)
    int32, new generator state).  m1/m2 (count,2) float32 enable the collinearity re-draw of the F estimator."""
    lib = _lib.load()
    st = ctypes.c_uint64(rng_state & 0xFFFFFFFFFFFFFFFF)
    out = np.empty((int(n_subsets), int(model_points)), np.int32)
    if m1 is not None:
        m1 = np.ascontiguousarray(m1, np.float32).reshape(-1, 2)
        m2 = np.ascontiguousarray(m2, np.float32).reshape(-1, 2)
        if m1.shape[0] != count or m2.shape[0] != count:
            raise ValueError("ransac_draw_subsets: point sets must have `count` rows")
    check(lib.vo_ransac_draw_subsets(ctypes.byref(st), int(count), int(model_points), int(n_subsets),
                                     _ptr(m1) if m1 is not None else None, _ptr(m2) if m2 is not None else None,
                                     int(max_attempts), _ptr(out)), "vo_ransac_draw_subsets")
    return out, int(st.value)


def refine_pose(ctx: Context, X, x, K, rvec, t, mask=None, max_iters: int = 10):
    """Gauss-Newton refit of one pose (rotation vector, translation) on the correspondences selected by mask
    (None = all), one launch -> (rvec (3,), t (3,), cost)."""
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    start = np.ascontiguousarray(np.hstack([np.asarray(rvec, np.float64).ravel(), np.asarray(t, np.float64).ravel()]))
    if start.size != 6:
        raise ValueError("refine_pose: rvec and t must have 3 entries each")
    if _is_torch(X):
        X, x = X.contiguous(), x.contiguous()
        if mask is not None:
            mask = mask.contiguous()
        mem, n = VO_MEM_DEVICE, int(X.shape[0])
    else:
        X = np.ascontiguousarray(X, np.float32).reshape(-1, 3)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8).ravel()
        mem, n = VO_MEM_HOST, int(X.shape[0])
    if int(x.shape[0]) != n or (mask is not None and int(mask.shape[0]) != n):
        raise ValueError("refine_pose: X, x and mask differ in length")
    out = np.empty(6, np.float64)
    cost = ctypes.c_double()
    check(ctx.lib.vo_refine_pose(ctx.handle, _ptr(X), _ptr(x), n, _ptr(mask) if mask is not None else None,
                                 Kd.ctypes.data_as(_lib.c_f64p), start.ctypes.data_as(_lib.c_f64p), int(max_iters),
                                 out.ctypes.data_as(_lib.c_f64p), ctypes.byref(cost), mem), "vo_refine_pose")
    return out[:3].copy(), out[3:].copy(), cost.value


def solve_pnp_ransac(ctx: Context, X, x, K, thr: float, conf: float, max_iters: int, refine_iters: int = 10,
                     rng_state: Optional[int] = None):
    """Whole PnP-RANSAC call (P3P candidates, OpenCV's loop replayed, refit on the inliers) ->
    (Rt (3,4) f64 or None, inlier mask (n,) u8, number of inliers, rng state after the draws)."""
    Kh = _k_host(K)
    if _is_torch(X):
        import torch
        X = X.reshape(-1, 3).contiguous(); x = x.reshape(-1, 2).contiguous()
        mask = torch.zeros(X.shape[0], dtype=torch.uint8, device=X.device)
        mem = VO_MEM_DEVICE
    else:
        X = np.ascontiguousarray(X, np.float32).reshape(-1, 3); x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
        mask = np.zeros(X.shape[0], np.uint8)
        mem = VO_MEM_HOST
    n = int(X.shape[0])
    if int(x.shape[0]) != n:
        raise ValueError("solve_pnp_ransac: point sets differ in length")
    state = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF if rng_state is None else rng_state)
    Rt = np.zeros((3, 4))
    good = ctypes.c_int32()
    check(ctx.lib.vo_solve_pnp_ransac(ctx.handle, _ptr(X), _ptr(x), n, Kh.ctypes.data_as(_lib.c_f64p), float(thr), float(conf),
                                      int(max_iters), int(refine_iters), ctypes.byref(state), Rt.ctypes.data_as(_lib.c_f64p),
                                      _ptr(mask), ctypes.byref(good), mem), "vo_solve_pnp_ransac")
    return (Rt if good.value > 0 else None), mask, int(good.value), int(state.value)


def track_pnp(ctx: Context, prev: Pyramid, nxt: Pyramid, pts, landmarks, K, thr: float, conf: float, max_iters: int,
              refine_iters: int = 10, klt_max_iters: int = 10, klt_eps: float = 0.001, klt_min_eig: float = 1e-2,
              rng_state: Optional[int] = None, next_image=None):
    """One tracked frame in one call: klt_track from prev to nxt, the status == 1 filter and solve_pnp_ransac on the
    survivors, with one synchronisation.  pts (n,2), landmarks (n,3), row for row; pyramids of one image each ->
    (next_pts (n,2) f32, status (n,) u8, Rt (3,4) f64 or None, inlier mask (n,) u8 over the input rows,
    number of tracked rows, number of inliers, rng state after the draws).  numpy or torch, as the points are.
    next_image (H,W) uint8 of the same kind: nxt is built from it inside the call (as nxt.build would) instead of
    beforehand; a host image then shares the page-locked upload of the points and nothing waits for it alone."""
    Kh = _k_host(K)
    prm = KltParams(int(klt_max_iters), float(klt_eps), float(klt_min_eig))
    if prev.batch != 1 or nxt.batch != 1:
        raise ValueError("track_pnp: pyramids of one image each")
    if _is_torch(pts):
        import torch
        p = _device_points(pts, 1, "track_pnp: pts")[0]
        X = landmarks
        if not (_is_torch(X) and X.is_cuda and X.dtype == torch.float32):
            raise ValueError("track_pnp: landmarks must be a float32 GPU tensor like pts")
        X = X.reshape(-1, 3).contiguous()
        n = int(p.shape[0])
        out = torch.empty_like(p)
        st = torch.empty(n, dtype=torch.uint8, device=p.device)
        mask = torch.zeros(n, dtype=torch.uint8, device=p.device)
        mem = VO_MEM_DEVICE
    else:
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        X = np.ascontiguousarray(landmarks, np.float32).reshape(-1, 3)
        n = int(p.shape[0])
        out = np.zeros_like(p)
        st = np.zeros(n, np.uint8)
        mask = np.zeros(n, np.uint8)
        mem = VO_MEM_HOST
    if int(X.shape[0]) != n:
        raise ValueError("track_pnp: pts and landmarks differ in length")
    img_ptr, img_stride = None, 0
    if next_image is not None:
        if _is_torch(next_image) != _is_torch(pts):
            raise ValueError("track_pnp: next_image must be of the same kind (numpy / torch) as pts")
        if _is_torch(next_image):
            import torch
            img = next_image
            if not (img.is_cuda and img.dtype == torch.uint8 and img.dim() == 2 and img.stride(1) == 1):
                raise ValueError("track_pnp: next_image must be a (H,W) uint8 GPU tensor with unit column stride")
            img_stride = int(img.stride(0))
            nxt._keep = img
        else:
            img = np.ascontiguousarray(next_image, dtype=np.uint8)
            if img.ndim != 2:
                raise ValueError("track_pnp: next_image must be (H,W)")
            img_stride = int(img.strides[0])
        if tuple(img.shape) != (nxt.height, nxt.width):
            raise ValueError(f"track_pnp: next_image {tuple(img.shape)} vs pyramid {(nxt.height, nxt.width)}")
        img_ptr = _ptr(img)
    state = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF if rng_state is None else rng_state)
    Rt = np.zeros((3, 4))
    tracked, good = ctypes.c_int32(), ctypes.c_int32()
    check(ctx.lib.vo_track_pnp(ctx.handle, prev.handle, nxt.handle, img_ptr, img_stride, _ptr(p), _ptr(X), n, ctypes.byref(prm),
                               Kh.ctypes.data_as(_lib.c_f64p), float(thr), float(conf), int(max_iters), int(refine_iters),
                               ctypes.byref(state), _ptr(out), _ptr(st), Rt.ctypes.data_as(_lib.c_f64p), _ptr(mask),
                               ctypes.byref(tracked), ctypes.byref(good), mem), "vo_track_pnp")
    return out, st, (Rt if good.value > 0 else None), mask, int(tracked.value), int(good.value), int(state.value)


def track_pnp_path(ctx: Context) -> int:
    """How the context's last track_pnp ended: 0 no pose sought, 1 on the single synchronisation, 2 more samples
    were needed, 3 the device-side sampler or replay could not vouch for its result."""
    return int(ctx.lib.vo_track_pnp_path(ctx.handle))


class TrackPnpBatch:
    """The tracked frame of many streams in one launch chain (vo_track_pnp_batch): stream b tracks image b of `prev`
    into image b of `nxt`, filters by status and runs PnP-RANSAC with its refit on the survivors.

    begin(...) enqueues the chain and returns; finish() waits once, re-checks every stream's loop on the host and
    returns (next_pts (B,n,2), status (B,n), Rt (B,3,4) f64, inlier mask (B,n), tracked (B,) i32, inliers (B,) i32
    -- 0: no pose, that stream's Rt row is zero --, rng states after the draws (B,) u64, path (B,) i32).  numpy or
    torch, as the points are; with torch the three arrays stay on the GPU.  One batch per context in flight."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._out = None

    def begin(self, prev: Pyramid, nxt: Pyramid, pts, landmarks, K, thr: float, conf: float, max_iters: int,
              refine_iters: int = 10, klt_max_iters: int = 10, klt_eps: float = 0.001, klt_min_eig: float = 1e-2,
              rng_state=None, out=None) -> "TrackPnpBatch":
        Kh = _k_host(K)
        prm = KltParams(int(klt_max_iters), float(klt_eps), float(klt_min_eig))
        B = prev.batch
        if nxt.batch != B:
            raise ValueError("track_pnp_batch: pyramids of different batch sizes")
        if _is_torch(pts):
            import torch
            p = _device_points(pts, B, "track_pnp_batch: pts")
            X = landmarks
            if not (_is_torch(X) and X.is_cuda and X.dtype == torch.float32):
                raise ValueError("track_pnp_batch: landmarks must be a float32 GPU tensor like pts")
            X = X.reshape(B, -1, 3).contiguous()
            n = int(p.shape[1])
            if out is None:
                out = (torch.empty_like(p), torch.empty((B, n), dtype=torch.uint8, device=p.device),
                       torch.empty((B, n), dtype=torch.uint8, device=p.device))
            for t, dt, shape, what in ((out[0], torch.float32, (B, n, 2), "next_pts"), (out[1], torch.uint8, (B, n), "status"),
                                       (out[2], torch.uint8, (B, n), "inlier mask")):
                if not (_is_torch(t) and t.is_cuda and t.dtype == dt and tuple(t.shape) == shape and t.is_contiguous()):
                    raise ValueError(f"track_pnp_batch: out {what} must be a contiguous {dt} GPU tensor of shape {shape}")
            mem = VO_MEM_DEVICE
        else:
            p = np.ascontiguousarray(pts, np.float32).reshape(B, -1, 2)
            X = np.ascontiguousarray(landmarks, np.float32).reshape(B, -1, 3)
            n = int(p.shape[1])
            out = (np.zeros_like(p), np.zeros((B, n), np.uint8), np.zeros((B, n), np.uint8))
            mem = VO_MEM_HOST
        if int(X.shape[1]) != n:
            raise ValueError("track_pnp_batch: pts and landmarks differ in length")
        states = None
        if rng_state is not None:
            states = np.ascontiguousarray(rng_state, np.uint64).reshape(B)
        check(self.ctx.lib.vo_track_pnp_batch_begin(self.ctx.handle, prev.handle, nxt.handle, _ptr(p), _ptr(X), n, ctypes.byref(prm),
                                                    Kh.ctypes.data_as(_lib.c_f64p), float(thr), float(conf), int(max_iters),
                                                    int(refine_iters), _ptr(states), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), mem),
              "vo_track_pnp_batch_begin")
        self._out = (out, B, (p, X, states))      # inputs kept alive until finish
        return self

    def finish(self):
        if self._out is None:
            raise RuntimeError("track_pnp_batch: finish without begin")
        out, B, _ = self._out
        self._out = None
        Rt = np.zeros((B, 3, 4))
        tracked, good, path = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        states = np.zeros(B, np.uint64)
        check(self.ctx.lib.vo_track_pnp_batch_finish(self.ctx.handle, _ptr(states), Rt.ctypes.data_as(_lib.c_f64p), _ptr(tracked),
                                                     _ptr(good), _ptr(path)), "vo_track_pnp_batch_finish")
        return out[0], out[1], Rt, out[2], tracked, good, states, path


def track_pnp_batch(ctx: Context, prev: Pyramid, nxt: Pyramid, pts, landmarks, K, thr: float, conf: float, max_iters: int,
                    **kw):
    """begin + finish of TrackPnpBatch in one call (one synchronisation for all streams)."""
    return TrackPnpBatch(ctx).begin(prev, nxt, pts, landmarks, K, thr, conf, max_iters, **kw).finish()


def find_fundamental_ransac(ctx: Context, m1, m2, thr: float, conf: float, max_iters: int, rng_state: Optional[int] = None):
    """Whole fundamental-matrix RANSAC call (7-point candidates, OpenCV's loop replayed, no refit) ->
    (F (3,3) f64 or None, inlier mask (n,) u8, number of inliers, rng state after the draws).  Host arrays."""
    m1 = np.ascontiguousarray(m1, np.float32).reshape(-1, 2); m2 = np.ascontiguousarray(m2, np.float32).reshape(-1, 2)
    n = int(m1.shape[0])
    if int(m2.shape[0]) != n:
        raise ValueError("find_fundamental_ransac: point sets differ in length")
    state = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF if rng_state is None else rng_state)
    F = np.zeros((3, 3))
    mask = np.zeros(n, np.uint8)
    good = ctypes.c_int32()
    check(ctx.lib.vo_find_fundamental_ransac(ctx.handle, _ptr(m1), _ptr(m2), n, float(thr), float(conf), int(max_iters),
                                             ctypes.byref(state), F.ctypes.data_as(_lib.c_f64p), _ptr(mask),
                                             ctypes.byref(good)), "vo_find_fundamental_ransac")
    return (F if good.value > 0 else None), mask, int(good.value), int(state.value)
