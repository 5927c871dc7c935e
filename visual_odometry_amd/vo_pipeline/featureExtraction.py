"""FeatureExtractor -- API of the reference's vo_pipeline/featureExtraction.py:7-51."""
import enum
from typing import Any, List, Tuple

import numpy as np

from .. import hip
from . import _runtime


class ExtractorType(enum.Enum):
    SIFT = 0,


class FeatureExtractor:
    """Computes image features for a given ExtractorType."""

    def __init__(self, extractor_type: ExtractorType = ExtractorType.SIFT):
        self.extractor_type = extractor_type
        self.extractor = None

    def get_kp(self, img: np.ndarray) -> Tuple[List[Any], np.ndarray]:
        """SIFT detect+describe (featureExtraction.py:20-35) is outside the accelerated
        path: descriptors are not produced by this build."""
        raise NotImplementedError("SIFT descriptors are not part of the GPU front end; "
                                  "use FeatureExtractor.harris / corners() for keypoints")

    @staticmethod
    def corners(img: np.ndarray, max_corners: int = 2000, quality: float = 0.01, radius: int = 5,
                k: float = 0.04, shi_tomasi: bool = True) -> np.ndarray:
        """Strongest `max_corners` local maxima of the Shi-Tomasi (or Harris) response, (M,2) float32 (x,y)."""
        ctx = _runtime.context()
        xy, _, cnt = hip.detect_corners(ctx, np.ascontiguousarray(img, dtype=np.uint8), int(max_corners), float(quality),
                                        int(radius), k, hip.MIN_EIGEN if shi_tomasi else hip.HARRIS)
        return xy[0, :int(cnt[0])]

    @staticmethod
    def harris(img: np.ndarray) -> np.ndarray:
        """Harris corners as the reference computes them (featureExtraction.py:37-51) -> (M, 2) float32 (x, y):
        response with blockSize 2, ksize 3, k = 0.1; 3x3 dilation; threshold at 0.01 * max; 8-connected
        components of the mask; their centroids -- row 0 is the centroid of the background, as in the
        reference, whose connectedComponentsWithStats call returns label 0 first --; cornerSubPix with an
        11x11 window, at most 50 iterations, eps 1e-3.  Response, mask and refinement run on the GPU
        (vo_harris_mask, vo_corner_subpix); the labelling of the mask, a host step between them, uses
        scipy.ndimage.label, which numbers the components in raster order of their first pixel."""
        from scipy import ndimage
        ctx = _runtime.context()
        gray = np.ascontiguousarray(img, dtype=np.uint8)
        mask = hip.harris_mask(ctx, gray, 0.1, 0.01, hip.HARRIS)
        labels, n = ndimage.label(mask, structure=np.ones((3, 3), np.int32))
        # centroids from the foreground pixels only; the background's (label 0, first row as in the reference)
        # follows from the totals over the whole image
        h, w = gray.shape
        ys, xs = np.nonzero(labels)
        lab = labels[ys, xs]
        area = np.bincount(lab, minlength=n + 1).astype(np.float64)
        sx = np.bincount(lab, weights=xs, minlength=n + 1)
        sy = np.bincount(lab, weights=ys, minlength=n + 1)
        area[0] = h * w - ys.size
        sx[0] = w * (w - 1) / 2.0 * h - xs.sum(dtype=np.float64)
        sy[0] = h * (h - 1) / 2.0 * w - ys.sum(dtype=np.float64)
        centroids = np.stack([sx, sy], axis=1) / np.maximum(area, 1)[:, None]
        return hip.corner_subpix(ctx, gray, np.float32(centroids), (5, 5), 50, 0.001)

    @staticmethod
    def harris_maxima(img: np.ndarray) -> np.ndarray:
        """Integer positions of the 3x3 local maxima of the same Harris response above the same threshold,
        strongest first (at most 4096), all on the GPU -> (M, 2) float32 (x, y)."""
        ctx = _runtime.context()
        xy, _, cnt = hip.detect_corners(ctx, np.ascontiguousarray(img, dtype=np.uint8), 4096, 0.01, 1, 0.1, hip.HARRIS)
        return xy[0, :int(cnt[0])]
