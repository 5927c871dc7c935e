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
        """Harris corners with the reference's parameters (blockSize 2, ksize 3, k = 0.1,
        threshold 0.01 * max; featureExtraction.py:37-51) -> (M, 2) float32 (x, y).

        The response is cv.cornerHarris' arithmetic; where the reference dilates,
        labels connected components and refines their centroids with cornerSubPix
        (returning the background blob's centroid as row 0), this returns the integer
        positions of the 3x3 local maxima above the same threshold, strongest first
        (at most 4096)."""
        ctx = _runtime.context()
        xy, _, cnt = hip.detect_corners(ctx, np.ascontiguousarray(img, dtype=np.uint8), 4096, 0.01, 1, 0.1, hip.HARRIS)
        return xy[0, :int(cnt[0])]
