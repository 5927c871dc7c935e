"""PoseEstimation -- KLT tracking and PnP-RANSAC wrappers with the reference's API
(vo_pipeline/poseEstimation.py:18-133); the cv2 calls are replaced by libvo_hip."""
import enum
from typing import Optional, Tuple

import numpy as np

from .. import hip, params
from . import _runtime, ransac
from .featureExtraction import ExtractorType, FeatureExtractor
from .featureMatching import FeatureMatcher, MatcherType
from .trackPoints import TrackPoints


class AlgoMethod(enum.Enum):
    DEFAULT = 0,
    P3P = 1,
    AP3P = 2,


class PoseEstimation:
    def __init__(self,
                 K: np.ndarray,
                 first_kps: Optional[np.ndarray] = None,
                 use_KLT: bool = True,
                 algo_method_type: AlgoMethod = AlgoMethod.DEFAULT):
        self.K = K
        self.prev_kpts = first_kps
        self.algo_method_type = algo_method_type
        self.algo_method = None
        self.get_method()
        self.use_KLT = use_KLT
        self.matcher = FeatureMatcher(MatcherType.FLANN, k=2)
        self.KLT_tracker = TrackPoints(params.KLT_RADIUS, params.KLT_N_ITERS, params.KLT_LAMBDA)

    def get_method(self):
        # Every method funnels into the same minimal solver: P3P on three points, a fourth to pick the root.  For P3P and
        # AP3P that is what cv.solvePnPRansac samples too; for DEFAULT (SOLVEPNP_ITERATIVE, reference
        # poseEstimation.py:41-42) OpenCV draws five-point samples and solves them iteratively -- a deliberate
        # difference here, off the active path (ContinuousVO constructs its estimator with AlgoMethod.P3P).
        self.algo_method = {AlgoMethod.DEFAULT: "ITERATIVE", AlgoMethod.P3P: "P3P",
                            AlgoMethod.AP3P: "AP3P"}[self.algo_method_type]

    def update_prev_kpts(self, kps: np.ndarray) -> None:
        self.prev_kpts = kps

    def PnP(self, pointcloud: np.ndarray, img_key_points: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """
        :param pointcloud:      (N, 3) landmarks matched to the keypoints
        :param img_key_points:  (N, 2) keypoints in the current image
        :return                 (M (4,4) float64 world->camera, inliers (k,1) int32)
        """
        assert pointcloud.shape[0] == img_key_points.shape[0]
        success, R, t, inliers = ransac.solve_pnp_ransac(
            pointcloud, img_key_points, self.K,
            thr=params.PNP_RANSAC_REPROJ_THRESHOLD,
            conf=params.PNP_RANSAC_CONFIDENCE,
            max_iters=params.PNP_RANSAC_MAX_ITERS)
        assert success, "PNP RANSAC was not able to compute a pose from 2D - 3D correspondences"
        M = np.eye(4)
        M[0:3, 0:3] = R
        M[0:3, 3] = np.ravel(t)
        return M, inliers

    def track_and_pose(self, img0: np.ndarray, img1: np.ndarray, prev_kpts: np.ndarray, pointcloud: np.ndarray,
                       reuse_pyramids: bool = False, want_inliers: bool = True):
        """KLT(img0, img1, prev_kpts), the status == 1 filter and PnP on the surviving rows -- the three steps of
        ContinuousVO._process_frame (reference continuousVO.py:160-180) -- in one library call with one
        synchronisation; same results as the separate calls.

        :param prev_kpts:   (N, 2) keypoints in img0
        :param pointcloud:  (N, 3) their landmarks, row for row
        :return (tracked_pts (N,2) float32, status (N,1) uint8, M (4,4) float64 or None when fewer than four
                 points survive or no sample gives a pose, inliers (k,1) int32 indexing the SURVIVING rows,
                 as PnP returns them -- None with want_inliers=False, for a caller that reads self.last_inlier_mask
                 and self.last_inlier_count instead)
        """
        a0 = np.ascontiguousarray(img0, dtype=np.uint8)
        a1 = np.ascontiguousarray(img1, dtype=np.uint8)
        if a0.ndim != 2 or a0.shape != a1.shape:
            raise ValueError("track_and_pose: img0 and img1 must be grayscale images of one size")
        pts = np.asarray(prev_kpts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        assert pointcloud.shape[0] == n
        self.last_inlier_count = 0
        if n == 0:
            self.last_inlier_mask = np.zeros(0, np.uint8)
            return np.zeros((0, 2), np.float32), np.zeros((0, 1), np.uint8), None, np.zeros((0, 1), np.int32)
        ctx = _runtime.context()
        # the new frame's pyramid is built inside the call, its upload riding with the points'
        p0, p1 = _runtime.pyramids_for_pair(a0, a1, params.KLT_NUM_PYRAMIDS, params.KLT_RADIUS, reuse=reuse_pyramids,
                                            defer_next=True)
        try:
            out, st, Rt, mask, _, self.last_inlier_count, _ = hip.track_pnp(
                ctx, p0, p1, pts, pointcloud, self.K, thr=params.PNP_RANSAC_REPROJ_THRESHOLD, conf=params.PNP_RANSAC_CONFIDENCE,
                max_iters=params.PNP_RANSAC_MAX_ITERS, refine_iters=10, klt_max_iters=params.KLT_N_ITERS, klt_eps=0.001,
                klt_min_eig=params.KLT_MIN_EIGEN_THRESHOLD, next_image=a1)
        except Exception:
            _runtime.forget_pyramids()          # the deferred pyramid may not have been built
            raise
        self.last_inlier_mask = mask        # over the INPUT rows (0 where a row was not tracked): what a caller that
                                            # keeps its own row bookkeeping reads instead of re-deriving it from `inliers`
        if Rt is None:
            return out, st.reshape(n, 1), None, np.zeros((0, 1), np.int32)
        M = np.eye(4)
        M[0:3, :] = Rt
        inliers = np.nonzero(mask[st == 1])[0].astype(np.int32).reshape(-1, 1) if want_inliers else None
        return out, st.reshape(n, 1), M, inliers

    def match_key_points(self, pointcloud: np.ndarray, img0: np.ndarray, img1: np.ndarray) -> np.ndarray:
        """use_KLT: track self.prev_kpts from img0 to img1 and keep the found ones.  Otherwise the
        reference's descriptor branch (poseEstimation.py:96-112): SIFT keypoints and descriptors of both images
        (FeatureExtractor.get_kp, on the GPU) matched on the GPU; the matched keypoints of img1 come back."""
        if self.use_KLT:
            pts1, st = self.KLT(img0, img1, self.prev_kpts)
            found = st == 1
            pts1 = pts1[found[:, 0]]
            self.update_prev_kpts(pts1)
        else:
            descriptor = FeatureExtractor(ExtractorType.SIFT)
            kp0, des0 = descriptor.get_kp(img0)
            kp1, des1 = descriptor.get_kp(img1)
            matches = self.matcher.match_descriptors(des0, des1)
            pts1 = np.zeros((len(matches), 2))
            for i, match in enumerate(matches):
                pts1[i, :] = kp1[match.trainIdx].pt
            self.update_prev_kpts(pts1)
        return pts1

    @staticmethod
    def KLT(img0: np.ndarray, img1: np.ndarray, prev_kpts: np.ndarray,
            reuse_pyramids: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """Pyramidal LK from img0 to img1 -> (tracked_pts (N,2) float32, status (N,1) uint8).

        reuse_pyramids (not in the reference, off by default): when img0 is the same array object that was
        img1 of the previous call and the caller has not written to it since, its pyramid is kept instead of
        being uploaded and rebuilt -- what a frame loop that tracks frame t-1 -> t, then t -> t+1 wants."""
        a0 = np.ascontiguousarray(img0, dtype=np.uint8)
        a1 = np.ascontiguousarray(img1, dtype=np.uint8)
        if a0.ndim != 2 or a0.shape != a1.shape:
            raise ValueError("KLT: img0 and img1 must be grayscale images of one size")
        pts = np.float32(prev_kpts).reshape(-1, 2)
        n = pts.shape[0]
        if n == 0:
            return np.zeros((0, 2), np.float32), np.zeros((0, 1), np.uint8)
        ctx = _runtime.context()
        # conversions above return the caller's own object when it already is contiguous uint8, so
        # object identity carries over from call to call
        p0, p1 = _runtime.pyramids_for_pair(a0, a1, params.KLT_NUM_PYRAMIDS, params.KLT_RADIUS, reuse=reuse_pyramids)
        out, st, _ = hip.klt_track(ctx, p0, p1, pts, max_iters=params.KLT_N_ITERS, eps=0.001,
                                   min_eig=params.KLT_MIN_EIGEN_THRESHOLD)
        return out[0], st[0].reshape(n, 1)
