"""Shared GPU contexts and pyramid cache of the drop-in classes."""
import concurrent.futures
import contextlib
import os
import threading
import weakref
from typing import Dict, Optional, Tuple

from .. import hip

_pyr_cache: Dict[Tuple[int, int, int, int, int, int], list] = {}
_side: Dict[int, hip.Context] = {}        # device -> the context on its own stream
_local = threading.local()                # .active: what context() hands out on this thread inside using() / side_context()


def context(device: int = 0) -> hip.Context:
    active = getattr(_local, "active", None)
    return active if active is not None else hip.default_context(device)


@contextlib.contextmanager
def using(ctx: hip.Context):
    """Inside the block every drop-in class called from this thread works on ctx."""
    previous = getattr(_local, "active", None)
    _local.active = ctx
    try:
        yield ctx
    finally:
        _local.active = previous


@contextlib.contextmanager
def side_context(device: int = 0):
    """Inside the block every drop-in class (called from this thread) works on a second context with a HIP stream of
    its own, so that its GPU work runs beside what the default context has queued instead of behind it
    (ContinuousVO builds a keyframe's two-view bootstrap there while the bundle-adjustment window's trial steps run).
    Host arrays in and out: nothing on the device is shared between the two contexts."""
    ctx = _side.get(device)
    if ctx is None:
        ctx = _side[device] = hip.Context.on_own_stream(device)
    with using(ctx):
        yield ctx


# (device, slot) -> (a context on a stream of its own, the one thread that works on it)
_beside: Dict[Tuple[int, int], Tuple[hip.Context, concurrent.futures.ThreadPoolExecutor]] = {}
helper_thread = os.environ.get("VO_HELPER_THREAD", "1") != "0"


def beside(fn, *args, device: Optional[int] = None, slot: int = 0, **kwargs) -> "concurrent.futures.Future":
    """fn(ctx, *args, **kwargs) on a helper thread, with a context (and HIP stream) that only that thread uses, while
    the caller goes on with work of its own: the library's calls release the interpreter lock, contexts share
    nothing, and calls that each alternate between short kernels and host work (two images' SIFT, the assembly of a
    bundle-adjustment window) fill each other's gaps.  `slot` names the helper: calls given to one slot run one after
    the other, different slots side by side.  The result comes from `.result()` of the returned future; device
    arrays in it belong to the helper's context and are complete when the call has returned (every library call ends
    synchronised).  device: the helper's GPU (default: that of the caller's active context)."""
    if device is None:
        device = context().device           # the GPU the caller works on
    entry = _beside.get((device, slot))
    if entry is None:
        entry = _beside[(device, slot)] = (hip.Context.on_own_stream(device),
                                           concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="vo-helper"))
    ctx, pool = entry
    return pool.submit(fn, ctx, *args, **kwargs)


def pyramid_pair(h: int, w: int, max_level: int, win: int = 21, device: int = 0):
    """Two reusable single-image pyramids for (h, w)."""
    ctx = context(device)
    key = (device, h, w, max_level, win, id(ctx))
    if key not in _pyr_cache:
        _pyr_cache[key] = [hip.Pyramid(ctx, 1, h, w, max_level, win) for _ in range(2)]
    return _pyr_cache[key]


# (device, h, w, max_level, win, context) -> (weak reference to the image whose pyramid sits in slot i, i)
_last_next: Dict[Tuple[int, int, int, int, int, int], Tuple["weakref.ref", int]] = {}


def forget_pyramids() -> None:
    """Drop the record of which frame's pyramid sits where (the next call rebuilds both)."""
    _last_next.clear()


def pyramids_for_pair(img0, img1, max_level: int, win: int = 21, device: int = 0, reuse: bool = False,
                      defer_next: bool = False):
    """Built pyramids (prev, next) of two frames of one size.  With reuse, a frame that is the very
    array object given as `img1` in the previous call keeps the pyramid built for it then (the caller
    vouches that it has not been written to since): only the new frame is uploaded and filtered.
    defer_next: the pyramid of img1 is left to the caller, who must build it into the returned object
    (hip.track_pnp does, given the image) before anything else uses it."""
    h, w = img0.shape
    key = (device, h, w, max_level, win, id(context(device)))
    pair = pyramid_pair(h, w, max_level, win, device)
    slot: Optional[int] = None
    if reuse and key in _last_next:
        ref, i = _last_next[key]
        if ref() is img0:
            slot = i
    if slot is None:
        slot = 0
        pair[0].build(img0)
    if not defer_next:
        pair[1 - slot].build(img1)
    try:
        _last_next[key] = (weakref.ref(img1), 1 - slot)
    except TypeError:           # not weak-referenceable (not an ndarray): nothing to remember
        _last_next.pop(key, None)
    return pair[slot], pair[1 - slot]
