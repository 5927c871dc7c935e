"""Shared GPU context and pyramid cache of the drop-in classes."""
from typing import Dict, Tuple

from .. import hip

_pyr_cache: Dict[Tuple[int, int, int, int, int], list] = {}


def context(device: int = 0) -> hip.Context:
    return hip.default_context(device)


def pyramid_pair(h: int, w: int, max_level: int, win: int = 21, device: int = 0):
    """Two reusable single-image pyramids for (h, w)."""
    key = (device, h, w, max_level, win)
    if key not in _pyr_cache:
        ctx = context(device)
        _pyr_cache[key] = [hip.Pyramid(ctx, 1, h, w, max_level, win) for _ in range(2)]
    return _pyr_cache[key]
