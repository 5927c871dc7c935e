"""BootstrapInitializer -- two-view initialisation with the reference's API and
attributes (vo_pipeline/bootstrap.py:13-261).

Differences that follow from the scope of this build: SIFT is not computed here.  By
default correspondences come from Shi-Tomasi corners tracked with the pyramidal LK
kernel (descriptors pts_des1/pts_des2 are then zero-filled (n,128) arrays); a caller
that has keypoints and descriptors of both images passes them as `features` and gets
the reference's route, two-nearest-neighbour matching with the ratio test
(bootstrap.py:188-211), on the GPU.  The fundamental matrix comes from
ransac.find_fundamental_mat (GPU-scored RANSAC) instead of cv2.findFundamentalMat.  The two-view geometry after F follows the
reference step by step (pinned by tests/golden/two_view.npz).
"""
import logging
from typing import Tuple

import numpy as np

from .. import hip, params
from . import _runtime, ransac
from .featureExtraction import FeatureExtractor
from .featureMatching import FeatureMatcher, MatcherType
from .poseEstimation import PoseEstimation


class BootstrapInitializer:
    def __init__(self, img1: np.ndarray, img2: np.ndarray, K: np.ndarray, max_point_dist: float = 100, features=None):
        """features: optional (kp0, des0, kp1, des1) -- keypoints of the two images as (n,2) arrays of (x, y) or
        sequences of objects with a `.pt`, and their (n,128) descriptors."""
        self.features = features
        self.img1 = img1
        self.img2 = img2
        self.K = K
        self.max_point_dist = max_point_dist
        self.kps = None
        F, pts1, pts2, pts_des1, pts_des2 = self._estimate_fundamental()
        self.F = F
        self.pts1, self.pts2 = pts1, pts2
        self.pts_des1, self.pts_des2 = pts_des1, pts_des2
        T, point_cloud, mask = self._transform_matrix(self.F, self.pts1, self.pts2)
        self.T = T
        self.point_cloud = point_cloud
        self.pts1 = self.pts1[mask]
        self.pts2 = self.pts2[mask]
        self.pts_des1 = self.pts_des1[mask]
        self.pts_des2 = self.pts_des2[mask]
        self.point_cloud = self.point_cloud[mask]

    @staticmethod
    def findFundamentalMat(pts0: np.ndarray, pts1: np.ndarray, ransacReprojThreshold: float, confidence: float,
                           maxIters: int):
        """(F (3,3) float64 or None, mask (n,1) uint8) -- the hook the reference fills with cv2.findFundamentalMat."""
        return ransac.find_fundamental_mat(pts0, pts1, ransacReprojThreshold, confidence, maxIters)

    @staticmethod
    def _keypoint_array(kp) -> np.ndarray:
        if isinstance(kp, np.ndarray):
            return np.asarray(kp, np.float32).reshape(-1, 2)
        return np.array([k.pt for k in kp], np.float32).reshape(-1, 2)

    def _matched_features(self):
        """Correspondences from descriptors as the reference forms them (bootstrap.py:188-211): best match of
        every descriptor of image 1 among those of image 2 that passes the ratio test."""
        kp0, des0, kp1, des1 = self.features
        kp0, kp1 = self._keypoint_array(kp0), self._keypoint_array(kp1)
        self.kps = kp1
        matcher = FeatureMatcher(MatcherType.FLANN, k=2, matching_threshold=params.MATCHING_THRESHOLD)
        idx, _, keep = matcher.match_arrays(des0, des1)
        q = np.nonzero(keep)[0]
        t = idx[q, 0]
        logging.info(f"Bootstrapping features: ({len(des0)}, {len(des1)}), matches: {q.size}")
        return kp0[q], kp1[t], np.asarray(des0, np.float32)[q], np.asarray(des1, np.float32)[t]

    def _estimate_fundamental(self, normalize=False):
        if self.features is not None:
            kp0, kp1, des0, des1 = self._matched_features()
            return self._fundamental_from(kp0, kp1, des0, des1)
        kp0 = FeatureExtractor.corners(self.img1, max_corners=3000, quality=0.01, radius=5)
        kp1, st = PoseEstimation.KLT(self.img1, self.img2, kp0)
        ok = st.ravel() == 1
        kp0, kp1 = kp0[ok], kp1[ok]
        self.kps = kp1
        n = kp0.shape[0]
        return self._fundamental_from(kp0, kp1, np.zeros((n, 128), dtype=np.float32), np.zeros((n, 128), dtype=np.float32))

    def _fundamental_from(self, kp0, kp1, des0, des1):
        n = kp0.shape[0]
        pts0 = np.ones((n, 3), dtype=np.float32)
        pts1 = np.ones((n, 3), dtype=np.float32)
        pts0[:, 0:2] = kp0
        pts1[:, 0:2] = kp1
        F, mask = self.findFundamentalMat(pts0[:, 0:2], pts1[:, 0:2], params.RANSAC_REPROJ_THRESHOLD,
                                          params.RANSAC_CONFIDENCE, params.RANSAC_MAX_ITERS)
        if F is None:
            raise RuntimeError("bootstrap: no fundamental matrix found")
        keep = mask.ravel() == 1
        return F, pts0[keep], pts1[keep], des0[keep], des1[keep]

    def _transform_matrix(self, F: np.ndarray, pts1: np.ndarray, pts2: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """E = K^T F K -> (R, t) by the 4-way cheirality test on triangulated points.
        Returns (T (4,4) float32, point cloud (n,4) in camera 1, feasibility mask)."""
        E = self.K.T @ F @ self.K
        Wm = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
        U, _, VT = np.linalg.svd(E)
        rotations = [U @ Wm @ VT, U @ Wm.T @ VT]
        rotations = [-R if np.linalg.det(R) < 0 else R for R in rotations]
        u = U[:, -1][np.newaxis].T
        if not np.all(u == 0):
            u = u / np.linalg.norm(u)
        M1 = self.K @ np.eye(3, 4, dtype=np.float32)
        best = None
        most = -np.inf
        for R_hat in rotations:
            for u_hat in (-u, u):
                T_hat = np.hstack((R_hat, u_hat))
                P_cam1 = self._linear_triangulation(pts1, pts2, M1, self.K @ T_hat)
                P_cam2 = (T_hat @ P_cam1.T).T
                in_front = np.sum(P_cam1[:, 2] > 0) + np.sum(P_cam2[:, 2] > 0)
                if in_front > most:
                    most = in_front
                    best = (R_hat, u_hat, P_cam1, -R_hat.T @ u_hat)
        R, t, point_cloud, cam2_W = best
        T = np.eye(4, dtype=np.float32)
        T[0:3, 0:3] = R
        T[0:3, 3] = t.ravel()
        min_real_z = min(0, cam2_W[2])
        close_enough = np.linalg.norm(point_cloud[:, 0:3], axis=1) <= self.max_point_dist
        in_front_cam = point_cloud[:, 2] > min_real_z
        mask = close_enough & in_front_cam
        logging.info(f"Rejected {np.sum(~mask)} points during 3D reconstruction.")
        return T, point_cloud, mask

    @staticmethod
    def _linear_triangulation(pts1: np.ndarray, pts2: np.ndarray, M1: np.ndarray, M2: np.ndarray) -> np.ndarray:
        """DLT triangulation of homogeneous pixel pairs (n,3) -> (n,4) float32 points with last
        column 1; one point per GPU lane instead of the reference's per-point SVD loop."""
        assert pts1.shape == pts2.shape, "The number of matched points in both images has to be the same"
        assert M1.shape == M2.shape == (3, 4), "Homogeneous projection matrices have be of shape (3,4)"
        return hip.triangulate(_runtime.context(), pts1, pts2, M1, M2)
