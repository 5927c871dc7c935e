"""FeatureExtractor -- API of the reference's vo_pipeline/featureExtraction.py:7-51."""
import enum
from typing import Any, List, Tuple

import numpy as np

from .. import hip
from . import _runtime


class ExtractorType(enum.Enum):
    SIFT = 0,


class KeyPoint:
    """The fields of cv2.KeyPoint that the pipeline reads (bootstrap.py:208-209 uses `.pt`)."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x: float, y: float, size: float, angle: float = -1.0, response: float = 0.0, octave: int = 0,
                 class_id: int = -1):
        self.pt = (float(x), float(y))
        self.size, self.angle, self.response = float(size), float(angle), float(response)
        self.octave, self.class_id = int(octave), int(class_id)

    def __repr__(self) -> str:
        return f"KeyPoint(pt={self.pt}, size={self.size:.3f}, angle={self.angle:.2f}, response={self.response:.4f}, octave={self.octave})"


class FeatureExtractor:
    """Computes image features for a given ExtractorType."""

    # the reference's SIFT arguments (featureExtraction.py:22-26)
    SIFT_ARGS = dict(n_octave_layers=4, contrast_threshold=0.03, edge_threshold=10.0, sigma=1.6)

    def __init__(self, extractor_type: ExtractorType = ExtractorType.SIFT):
        self.extractor_type = extractor_type
        self.extractor = None
        self.get_extractor()

    def get_extractor(self) -> None:
        if self.extractor_type == ExtractorType.SIFT:
            self.extractor = self.sift_arrays

    @classmethod
    def sift_arrays(cls, img: np.ndarray, device_descriptors: bool = False) -> Tuple[np.ndarray, Any]:
        """SIFT on the GPU -> (keypoints as a structured array (x, y, size, angle, response, octave), descriptors
        (n, 128) float32): the array form of get_kp, without one Python object per keypoint.
        device_descriptors: leave the descriptors on the GPU (a hip.DeviceArray, library-owned memory on the
        context's stream) for a matcher that takes them there -- 10 MB per 20 000 keypoints that need not cross the
        bus twice."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        return hip.sift_detect_compute(_runtime.context(), img, device_descriptors=device_descriptors, **cls.SIFT_ARGS)

    def get_kp(self, img: np.ndarray) -> Tuple[List[Any], np.ndarray]:
        """Keypoints and (num_keypoints, 128) float32 descriptors of img (featureExtraction.py:28-35), computed by
        vo_sift_detect_compute: the algorithm of cv.SIFT_create(0, 4, 0.03, 10, 1.6).detectAndCompute(img, None) as
        OpenCV 4.5.4 publishes it, bit-identical to this build's CPU restatement of it; against cv2 itself the
        result is unpinned (cv2 is not installed where this was built, and the reference ships no SIFT vectors)."""
        kp, des = self.extractor(img)
        return [KeyPoint(k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"]) for k in kp], des

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
        11x11 window, at most 50 iterations, eps 1e-3.  All of it is one library call (vo_harris_corners)
        with one synchronisation.  Rows follow the numbering of the block-based labeller of the OpenCV
        version the reference pins (components in raster order of their first 2x2 block); that order is
        unpinned against cv2 itself, which is not installed here."""
        ctx = _runtime.context()
        gray = np.ascontiguousarray(img, dtype=np.uint8)
        room = 4096                                  # a Harris mask rarely has more blobs; retry with full room if it does
        corners = hip.harris_corners(ctx, gray, 0.1, 0.01, hip.CC_FIRST_BLOCK, (5, 5), 50, 0.001, max_corners=room)
        if corners.shape[0] == room:
            corners = hip.harris_corners(ctx, gray, 0.1, 0.01, hip.CC_FIRST_BLOCK, (5, 5), 50, 0.001)
        return corners

    @staticmethod
    def harris_maxima(img: np.ndarray) -> np.ndarray:
        """Integer positions of the 3x3 local maxima of the same Harris response above the same threshold,
        strongest first (at most 4096), all on the GPU -> (M, 2) float32 (x, y)."""
        ctx = _runtime.context()
        xy, _, cnt = hip.detect_corners(ctx, np.ascontiguousarray(img, dtype=np.uint8), 4096, 0.01, 1, 0.1, hip.HARRIS)
        return xy[0, :int(cnt[0])]
