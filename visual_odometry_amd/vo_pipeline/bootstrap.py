"""BootstrapInitializer -- two-view initialisation with the reference's API and
attributes (vo_pipeline/bootstrap.py:13-261).

Correspondences come, as in the reference (bootstrap.py:179-211), from SIFT keypoints of both
images matched by two nearest neighbours with the ratio test, all on the GPU; a caller that
already has keypoints and descriptors passes them as `features`.  `correspondences="klt"`
selects the lighter route of the project statement instead: Shi-Tomasi corners tracked with the
pyramidal LK kernel (descriptors pts_des1/pts_des2 are then zero-filled (n,128) arrays).  The fundamental matrix comes from
ransac.find_fundamental_mat (GPU-scored RANSAC) instead of cv2.findFundamentalMat.  The two-view geometry after F (motion hypotheses, cheirality vote, feasibility mask) is this
build's own formulation of bootstrap.py:76-138 with the vote on the GPU; its outputs are pinned by
tests/golden/two_view.npz (outputs of the reference functions).
"""
import logging
from typing import Tuple

import numpy as np

from .. import hip, params
from . import _runtime, ransac
from .featureExtraction import FeatureExtractor
from .featureMatching import FeatureMatcher, MatcherType
from .poseEstimation import PoseEstimation


class _DescriptorRows:
    """Rows of a descriptor matrix picked by index; a matrix that lives on the GPU is read only when the rows are
    asked for (the pipeline itself never looks at the descriptors of a bootstrap's correspondences, and fetching
    them costs more than the RANSAC call they accompany)."""

    def __init__(self, source, rows):
        self.source, self.rows = source, np.asarray(rows)

    def __getitem__(self, selection):
        return _DescriptorRows(self.source, self.rows[selection])

    def numpy(self) -> np.ndarray:
        if isinstance(self.source, np.ndarray):
            return np.asarray(self.source, np.float32)[self.rows]
        if isinstance(self.source, hip.DeviceArray):
            return self.source.numpy()[self.rows]
        import torch
        rows = torch.from_numpy(np.ascontiguousarray(self.rows, dtype=np.int64)).to(self.source.device)
        return self.source[rows].cpu().numpy()


class _PendingSift:
    """SIFT features of two images being computed on helper threads (BootstrapInitializer.sift_features_begin)."""

    def __init__(self, first, second):
        self.calls = (first, second)

    def result(self):
        """(kp0 (n,2), des0, kp1 (m,2), des1), descriptors as device arrays; waits for both calls."""
        failed = [c.exception() for c in self.calls]        # waits: neither call is left running if one failed
        for f in failed:
            if f is not None:
                raise f
        (k0, d0), (k1, d1) = (c.result() for c in self.calls)
        return np.stack([k0["x"], k0["y"]], axis=1), d0, np.stack([k1["x"], k1["y"]], axis=1), d1

    def discard(self) -> None:
        """Waits for the calls and drops what they return (the caller failed meanwhile)."""
        for c in self.calls:
            if c.exception() is None:
                des = c.result()[1]
                if des is not None:
                    des.close()


class BootstrapInitializer:
    def __init__(self, img1: np.ndarray, img2: np.ndarray, K: np.ndarray, max_point_dist: float = 100, features=None,
                 correspondences: str = "sift"):
        """features: optional (kp0, des0, kp1, des1) -- keypoints of the two images as (n,2) arrays of (x, y) or
        sequences of objects with a `.pt`, and their (n,128) descriptors -- or what sift_features_begin returned.  correspondences: "sift" (the reference's
        route) or "klt" (corners + tracking), used when no features are given."""
        if correspondences not in ("sift", "klt"):
            raise ValueError(f"BootstrapInitializer: correspondences must be 'sift' or 'klt', got {correspondences!r}")
        self.features = features
        self.correspondences = correspondences
        self.img1 = img1
        self.img2 = img2
        self.K = K
        self.max_point_dist = max_point_dist
        self.kps = None
        F, pts1, pts2, pts_des1, pts_des2 = self._estimate_fundamental()
        self.F = F
        self.pts1, self.pts2 = pts1, pts2
        T, point_cloud, mask = self._transform_matrix(self.F, self.pts1, self.pts2)
        self.T = T
        self.point_cloud = point_cloud
        self.pts1 = self.pts1[mask]
        self.pts2 = self.pts2[mask]
        self._des1, self._des2 = pts_des1[mask], pts_des2[mask]
        self.point_cloud = self.point_cloud[mask]

    # descriptors of the surviving correspondences, (n,128) float32 as in the reference (bootstrap.py:49-50)
    @property
    def pts_des1(self) -> np.ndarray:
        if isinstance(self._des1, _DescriptorRows):
            self._des1 = self._des1.numpy()
        return self._des1

    @pts_des1.setter
    def pts_des1(self, value):
        self._des1 = value

    @property
    def pts_des2(self) -> np.ndarray:
        if isinstance(self._des2, _DescriptorRows):
            self._des2 = self._des2.numpy()
        return self._des2

    @pts_des2.setter
    def pts_des2(self, value):
        self._des2 = value

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
        on_torch = not isinstance(des0, (np.ndarray, hip.DeviceArray)) and hasattr(des0, "is_cuda")
        if on_torch:
            # the caller's tensors were written on torch's current stream, the library works on its context's stream:
            # order the two by waiting on both sides of the call
            import torch
            torch.cuda.current_stream(des0.device).synchronize()
        idx, _, keep = matcher.match_arrays(des0, des1)
        if on_torch:
            _runtime.context().sync()
        if isinstance(des0, hip.DeviceArray):
            # descriptors in library-owned device memory: the (small) match arrays are numpy, the rows stay where they are
            q = np.nonzero(keep)[0]
            t = idx[q, 0]
            logging.info(f"Bootstrapping features: ({len(des0)}, {len(des1)}), matches: {q.size}")
            return kp0[q], kp1[t], _DescriptorRows(des0, q), _DescriptorRows(des1, t)
        if not isinstance(des0, np.ndarray) and hasattr(des0, "is_cuda"):
            # descriptors on the GPU: only the matched rows come back
            import torch
            qd = torch.nonzero(keep).ravel()
            td = idx[qd, 0].long()
            q, t = qd.cpu().numpy(), td.cpu().numpy()
            logging.info(f"Bootstrapping features: ({len(des0)}, {len(des1)}), matches: {q.size}")
            return kp0[q], kp1[t], _DescriptorRows(des0, q), _DescriptorRows(des1, t)
        q = np.nonzero(keep)[0]
        t = idx[q, 0]
        logging.info(f"Bootstrapping features: ({len(des0)}, {len(des1)}), matches: {q.size}")
        return kp0[q], kp1[t], np.asarray(des0, np.float32)[q], np.asarray(des1, np.float32)[t]

    @staticmethod
    def sift_features_begin(img1: np.ndarray, img2: np.ndarray) -> _PendingSift:
        """Starts the SIFT calls of a bootstrap's two images on two helper threads (contexts of their own) and
        returns at once; hand the result to the constructor as `features`.  For a caller that has other work to do
        before it needs the reconstruction -- ContinuousVO starts them before it assembles a keyframe's
        bundle-adjustment window."""
        calls = [_runtime.beside(hip.sift_detect_compute, np.ascontiguousarray(img, dtype=np.uint8), slot=slot,
                                 device_descriptors=True, **FeatureExtractor.SIFT_ARGS) for slot, img in enumerate((img1, img2))]
        return _PendingSift(*calls)

    def _estimate_fundamental(self, normalize=False):
        if isinstance(self.features, _PendingSift):
            self.features = self.features.result()
        if self.features is None and self.correspondences == "sift":
            if _runtime.helper_thread:
                # the second image's SIFT on a helper thread and a stream of its own, beside the first image's
                second = _runtime.beside(hip.sift_detect_compute, np.ascontiguousarray(self.img2, dtype=np.uint8),
                                         device_descriptors=True, **FeatureExtractor.SIFT_ARGS)
                try:
                    k0, d0 = FeatureExtractor.sift_arrays(self.img1, device_descriptors=True)
                finally:
                    done = second.exception()       # waits: nothing of the helper's call is left running either way
                if done is not None:
                    raise done
                k1, d1 = second.result()
            else:
                k0, d0 = FeatureExtractor.sift_arrays(self.img1, device_descriptors=True)
                k1, d1 = FeatureExtractor.sift_arrays(self.img2, device_descriptors=True)
            self.features = (np.stack([k0["x"], k0["y"]], axis=1), d0, np.stack([k1["x"], k1["y"]], axis=1), d1)
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

    @staticmethod
    def motion_hypotheses(E: np.ndarray) -> np.ndarray:
        """The four motions [R|t] (4,3,4) float64 an essential matrix admits, in the order the vote
        visits them (bootstrap.py:89-111): two rotations from the SVD of E, each paired with the
        unit left-null direction of E taken negative first."""
        U, _, Vt = np.linalg.svd(E)
        quarter_turn = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
        baseline = U[:, 2].copy()
        if np.any(baseline != 0):
            baseline /= np.linalg.norm(baseline)
        motions = np.empty((4, 3, 4))
        for r, twist in enumerate((quarter_turn, quarter_turn.T)):
            R = U @ twist @ Vt
            if np.linalg.det(R) < 0:         # a reflection: E is defined up to sign
                R = -R
            for s, sign in enumerate((-1.0, 1.0)):
                motions[2 * r + s, :, :3] = R
                motions[2 * r + s, :, 3] = sign * baseline
        return motions

    def _transform_matrix(self, F: np.ndarray, pts1: np.ndarray, pts2: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Relative pose from F (bootstrap.py:76-138): E = K^T F K, its four motion hypotheses put to the
        vote of points in front of both cameras -- all four triangulations and the four counts in one GPU
        launch --, then the feasibility mask on the winner's cloud.
        Returns (T (4,4) float32, point cloud (n,4) float32 in camera 1, mask (n,) bool)."""
        motions = self.motion_hypotheses(self.K.T @ F @ self.K)
        M1 = self.K @ np.eye(3, 4, dtype=np.float32)
        M2 = np.stack([self.K @ m for m in motions])
        clouds, votes = hip.cheirality_vote(_runtime.context(), pts1, pts2, M1, M2, motions)
        won = int(np.argmax(votes))              # first of equal counts, like a strict '>' scan
        point_cloud = clouds[won]
        T = np.eye(4, dtype=np.float32)
        T[0:3, :] = motions[won]
        # camera 2's centre in camera-1 coordinates; points may lie behind camera 1 only as far as it does
        centre2_z = float(-(motions[won, :, :3].T @ motions[won, :, 3])[2])
        depth_floor = min(0.0, centre2_z)
        mask = (np.linalg.norm(point_cloud[:, 0:3], axis=1) <= self.max_point_dist) & (point_cloud[:, 2] > depth_floor)
        logging.info("two-view reconstruction: %d of %d points rejected (too far or behind the cameras)",
                     int(np.count_nonzero(~mask)), mask.size)
        return T, point_cloud, mask

    @staticmethod
    def _linear_triangulation(pts1: np.ndarray, pts2: np.ndarray, M1: np.ndarray, M2: np.ndarray) -> np.ndarray:
        """DLT triangulation of homogeneous pixel pairs (n,3) -> (n,4) float32 points with last
        column 1; one point per GPU lane instead of the reference's per-point SVD loop."""
        assert pts1.shape == pts2.shape, f"point sets differ: {pts1.shape} vs {pts2.shape}"
        assert M1.shape == (3, 4) and M2.shape == (3, 4), "projection matrices must be (3,4)"
        return hip.triangulate(_runtime.context(), pts1, pts2, M1, M2)
