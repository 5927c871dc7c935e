"""TrackPoints -- KLT tracking with the bidirectional (forward-backward) test.

Mirrors the API of the reference's vo_pipeline/trackPoints.py:12-161.  The
reference body is a non-functional draft (trackKLT raises on the first accepted
keypoint and is never called); what is kept is its interface and its intent:
track img0 -> img1, track the result back, accept when
|kp - kp_back| < lambda_threshold in both axes (trackPoints.py:54-61, 154-161).
Both passes run the pyramidal LK kernel the pipeline's working tracker uses
(PoseEstimation.KLT).
"""
import numpy as np

from .. import hip, params
from . import _runtime


class TrackPoints:

    def __init__(self, patch_radius: float, num_iters_GN: float, lambda_threshold: float):
        self.patch_radius = patch_radius
        self.num_iters_GN = num_iters_GN
        self.lambda_threshold = lambda_threshold

    def trackKLT(self, img0: np.ndarray, img1: np.ndarray, keypoints: np.ndarray):
        """
        :param img0:        reference image the keypoints live in
        :param img1:        image to track the keypoints into
        :param keypoints:   (N, 2) keypoints (x, y)
        :return             (keypoints1 (N, 2) float64 -- zero rows where rejected,
                             pointcloud_mask (N, 1) bool)
        """
        img0 = np.ascontiguousarray(img0, dtype=np.uint8)
        img1 = np.ascontiguousarray(img1, dtype=np.uint8)
        if img0.shape != img1.shape or img0.ndim != 2:
            raise ValueError("trackKLT: img0 and img1 must be grayscale images of one size")
        kps = np.asarray(keypoints, dtype=np.float32).reshape(-1, 2)
        n = kps.shape[0]
        if n == 0:
            return np.zeros((0, 2)), np.zeros((0, 1), dtype=bool)
        ctx = _runtime.context()
        p0, p1 = _runtime.pyramid_pair(img0.shape[0], img0.shape[1], params.KLT_NUM_PYRAMIDS, int(self.patch_radius))
        p0.build(img0)
        p1.build(img1)
        out, mask = hip.klt_track_fb(ctx, p0, p1, kps, lam=float(self.lambda_threshold),
                                     max_iters=int(self.num_iters_GN), eps=0.001,
                                     min_eig=params.KLT_MIN_EIGEN_THRESHOLD)
        return out[0].astype(np.float64), mask[0].astype(bool).reshape(n, 1)
