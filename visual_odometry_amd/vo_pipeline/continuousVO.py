"""ContinuousVO -- the caller of the hot path (reference: vo_pipeline/continuousVO.py:19-352):
frame loop, keyframe decision, windowed bundle adjustment and re-bootstrap, plus the
marshalling helpers that shape the arrays the kernels consume.  Every numeric operator it
calls (KLT, PnP-RANSAC, two-view bootstrap, bundle adjustment) is the GPU drop-in of this
package; the bookkeeping between them is numpy on whole frames."""
import logging
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import hip, params
from ..utils.frameQueue import FrameQueue
from ..utils.matrix import hom_inv
from . import _runtime
from .bootstrap import BootstrapInitializer
from .bundleAdjustment import BundleAdjustment
from .featureExtraction import ExtractorType, FeatureExtractor
from .frameState import FrameState
from .keypointTrajectory import KeypointTrajectories
from .poseEstimation import AlgoMethod, PoseEstimation

log = logging.getLogger(__name__)


def filter_tracked(status: np.ndarray, tracked_pts: np.ndarray, *per_point: np.ndarray):
    """Keep rows with status == 1 in their original order (continuousVO.py:168-172)."""
    keep = np.asarray(status).ravel() == 1
    return (tracked_pts[keep],) + tuple(a[keep] for a in per_point)


def inlier_ratio(inliers: np.ndarray, inlier_count: int) -> float:
    """len(inliers) / inlier_count with the count fixed at the last bootstrap (continuousVO.py:178)."""
    return inliers.shape[0] / inlier_count


def baseline_uncertainty(T0: np.ndarray, T1: np.ndarray, landmarks: np.ndarray, keep: Optional[np.ndarray] = None) -> float:
    """|camera-centre displacement| / mean positive landmark depth in camera 0 (continuousVO.py:249-271);
    inf when no landmark has positive depth.  keep (not in the reference): a boolean mask over the rows of
    landmarks -- the depths of landmarks[keep] without building that array."""
    # the depth of X in camera 0 is the third row of its pose applied to X -- the reference's (X - C0) . (R0^T e3) with
    # the centre C0 = -R0^T t0 multiplied out --, and the centres' distance needs no full inverse either: this runs
    # on every tracked frame
    T0, T1 = np.asarray(T0, np.float64), np.asarray(T1, np.float64)
    d = landmarks @ T0[2, 0:3] + T0[2, 3]
    d = d[d > 0] if keep is None else d[keep & (d > 0)]
    if d.size == 0:
        return np.inf
    between = T0[0:3, 0:3].T @ T0[0:3, 3] - T1[0:3, 0:3].T @ T1[0:3, 3]
    return float(np.sqrt(between @ between) / d.mean())


def _tracked_frame_rows(status: np.ndarray, inlier_mask: np.ndarray):
    """One pass over what the fused tracked-frame call returns: (rows with status 1 as a boolean mask, their number,
    the INPUT rows of the PnP inliers in ascending order).  Equals filter_tracked followed by np.unique(inliers) and the
    gathers through them, without building the intermediate arrays."""
    keep = np.asarray(status).ravel() == 1
    return keep, int(np.count_nonzero(keep)), np.flatnonzero(inlier_mask)


def assemble_ba_window(poses: Sequence[np.ndarray], per_frame: Sequence[Tuple[np.ndarray, np.ndarray]],
                       landmark_of_traj: Dict[int, np.ndarray]):
    """Build the BA arrays for a window of frames (continuousVO.py:287-313).

    poses[i]: (4,4) or (3,4) pose of window frame i; per_frame[i] = (keypoints (N_i,2), traj ids (N_i,));
    landmark_of_traj: trajectory id -> (3,) landmark.
    -> (poses (F,3,4) f32, landmarks (P,3) f32, camera_indices (O,) int64 ascending,
        point_indices (O,) int64, keypoints (O,2) f32, traj ids in landmark order)."""
    traj2local: Dict[int, int] = {}
    lm: List[np.ndarray] = []
    ci: List[int] = []
    pi: List[int] = []
    kps: List[np.ndarray] = []
    for i, (kp, traj) in enumerate(per_frame):
        for k, t in zip(kp, traj):
            t = int(t)
            if t not in traj2local:
                traj2local[t] = len(lm)
                lm.append(np.asarray(landmark_of_traj[t]))
            pi.append(traj2local[t])
            ci.append(i)
            kps.append(k)
    P = np.asarray([np.asarray(p)[0:3, 0:4] for p in poses], dtype=np.float32)
    order = sorted(traj2local, key=traj2local.get)
    return (P, np.asarray(lm, dtype=np.float32).reshape(-1, 3), np.asarray(ci, dtype=np.int64),
            np.asarray(pi, dtype=np.int64), np.asarray(kps, dtype=np.float32).reshape(-1, 2), order)


def window_indices(traj_per_frame: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Observation indexing of a BA window (continuousVO.py:287-306) from the trajectory ids seen on
    each window frame -> (camera_indices (O,) ascending, point_indices (O,), trajectory id of each
    local landmark in order of first appearance)."""
    counts = [len(t) for t in traj_per_frame]
    ci = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    if ci.size == 0:
        return ci, np.zeros(0, np.int64), np.zeros(0, np.int64)
    flat = np.concatenate([np.asarray(t, np.int64) for t in traj_per_frame])
    lo = int(flat.min())
    span = int(flat.max()) - lo + 1
    if lo < 0 or span > 8 * flat.size + 1024:           # sparse ids: sort-based
        uniq, first, inv = np.unique(flat, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        return ci, rank[inv].astype(np.int64), uniq[order]
    # dense ids (trajectory ids are consecutive integers): a table instead of a sort over all observations.  Written
    # back to front, an id's slot ends up holding the position of its first occurrence.
    first = np.full(span, -1, np.int64)
    rel = flat - lo
    first[rel[::-1]] = np.arange(flat.size - 1, -1, -1, dtype=np.int64)
    ids = np.nonzero(first >= 0)[0]
    ids = ids[np.argsort(first[ids], kind="stable")]     # order of first appearance
    rank = np.empty(span, np.int64)
    rank[ids] = np.arange(ids.size, dtype=np.int64)
    return ci, rank[rel], ids + lo


class ContinuousVO:
    """Same constructor, attributes and step() protocol as the reference.  Bootstraps take their
    correspondences from SIFT + two-nearest-neighbour matching like the reference
    (bootstrap_correspondences="sift"), or from Shi-Tomasi corners tracked by KLT ("klt").

    rescale_bootstrap: a re-bootstrap triangulates with a unit baseline (bootstrap.py:103-111).
    False (default) is the reference's active code path: every keyframe restarts the scale.  True is
    an opt-in extension: the new translation and landmarks are scaled so that the baseline equals
    the one implied by the tracked pose -- the step the reference carries as commented-out code
    (continuousVO.py:128-135) -- which keeps one world scale across keyframes.

    fused_step: a tracked frame's KLT, status filter and PnP run as one library call with one
    synchronisation (PoseEstimation.track_and_pose) instead of the reference's two calls with the filter
    between them; the results are identical, False takes the two calls.

    overlap_keyframe: at a keyframe the reference adjusts the window and then bootstraps anew, one after the
    other (continuousVO.py:223-236).  The GPU part of the bootstrap -- features or corners of the two views, their
    correspondences, the fundamental matrix, the cheirality vote -- reads neither the window's result nor the
    track store, so with True (default) it runs on a second context while the window's trial steps are in
    flight on the first, and the host waits for the window only then; everything that touches poses, landmarks
    and trajectories happens in the reference's order afterwards.  The results are those of the sequential
    order (False)."""

    def __init__(self, dataset, descriptorType=ExtractorType.SIFT, matcherType=None, useKLT=True,
                 algo_method=AlgoMethod.P3P, max_point_distance: int = 100, frames_to_skip: int = 10,
                 frame_queue_size: int = 1000, rescale_bootstrap: bool = False,
                 bootstrap_correspondences: str = "sift", fused_step: bool = True, overlap_keyframe: bool = True) -> None:
        self.dataset = dataset
        self.descriptor = FeatureExtractor(descriptorType)
        self.matcher = matcherType
        self.useKLT = useKLT
        self.poseEstimator = PoseEstimation(self.dataset.K, use_KLT=useKLT, algo_method_type=algo_method)
        self.bundle_adjustment = BundleAdjustment(self.dataset.K)
        assert frame_queue_size > 0
        self.frame_queue = FrameQueue(frame_queue_size)
        self.keyframes: List[FrameState] = []
        self.K: Optional[np.ndarray] = None
        self.max_point_distance = max_point_distance
        self.frames_to_skip = frames_to_skip
        self.rescale_bootstrap = rescale_bootstrap
        self.bootstrap_correspondences = bootstrap_correspondences
        self.fused_step = fused_step
        self.overlap_keyframe = overlap_keyframe
        self.keypoint_trajectories = KeypointTrajectories()
        self.frame_idx = 0
        self.bootstrap_idx: List[int] = []
        self.inlier_count = 0

    def step(self) -> Optional[FrameState]:
        """Take the next frame: the first `frames_to_skip` frames are only queued, the next one
        initialises the point cloud against the oldest queued frame, every later one is tracked.
        Returns the newest FrameState, or None when the dataset is exhausted."""
        try:
            K, img = next(self.dataset.frames)
        except StopIteration:
            return None
        if self.frame_idx < self.frames_to_skip:
            self.K = K
            self.frame_queue.add(FrameState(self.frame_idx, img, np.eye(4, dtype=np.float32)))
        elif self.frame_idx == self.frames_to_skip:
            self._init_bootstrap(self.frame_idx, img)
        else:
            self._process_frame(self.frame_idx, img)
        self.frame_idx += 1
        return self.frame_queue.get_head()

    def _add_keyframe(self, frame_state: FrameState) -> None:
        self.keyframes.append(frame_state)

    def _init_bootstrap(self, idx: int, img: np.ndarray) -> None:
        baseline = self.frame_queue.get(self.frames_to_skip - 1)
        views = None
        if self.overlap_keyframe:
            # Set-up belongs to the first bootstrap, not to the first keyframe of the loop: done the way the keyframes
            # do theirs -- feature calls on the helper threads, the rest on the second context --, it creates those
            # contexts and sizes their buffers (page-locked rings, staging, scale spaces: several milliseconds of
            # allocation calls); the tracker's pyramids and the solver's pool on the first context follow.
            features = None
            if self.bootstrap_correspondences == "sift" and _runtime.helper_thread:
                features = BootstrapInitializer.sift_features_begin(baseline.img, img)
            with _runtime.side_context():
                views = self._bootstrap_views(baseline, img, features)
            self._reserve(img)
        T = self._bootstrap(baseline, idx, img, views=views)
        frame_state = FrameState(idx, img, T, is_key=True)
        self.frame_queue.add(frame_state)
        self._add_keyframe(frame_state)

    def _reserve(self, img: np.ndarray) -> None:
        """What the loop will need on the first context, allocated now: the tracker's pyramids for this image size and
        the buffers of a tracked frame (eight points of the image tracked onto itself; the image's pyramid is then the
        one the next frame starts from), and the bundle-adjustment pool (a one-observation solver, created and
        closed, leaves it behind)."""
        h, w = img.shape[:2]
        corner = np.array([[w / 4, h / 4], [w / 2, h / 4], [3 * w / 4, h / 4], [w / 4, h / 2], [3 * w / 4, h / 2],
                           [w / 4, 3 * h / 4], [w / 2, 3 * h / 4], [3 * w / 4, 3 * h / 4]], np.float32)
        depth = np.hstack([corner, np.full((8, 1), 10.0, np.float32)])
        if self.fused_step:
            self.poseEstimator.track_and_pose(img, img, corner, depth, reuse_pyramids=True)
        else:
            PoseEstimation.KLT(img, img, corner, reuse_pyramids=True)
        one = np.zeros(1, np.int64)
        hip.BaSolver(_runtime.context(), 1, 1, one, one, np.zeros((1, 2)), np.asarray(self.K, np.float64), 0).close()

    def _bootstrap_views(self, baseline: FrameState, img: np.ndarray, features=None) -> BootstrapInitializer:
        """The two-view reconstruction of (baseline, img): images in, relative pose and cloud out; it reads no pose and
        nothing of the track store.  features: the two images' SIFT features already asked for
        (BootstrapInitializer.sift_features_begin)."""
        return BootstrapInitializer(baseline.img, img, self.K, max_point_dist=self.max_point_distance, features=features,
                                    correspondences=self.bootstrap_correspondences)

    def _bootstrap(self, baseline: FrameState, frame_idx: int, img: np.ndarray,
                   world_transform: Optional[np.ndarray] = None, views: Optional[BootstrapInitializer] = None) -> np.ndarray:
        """Two-view initialisation between baseline and img (`views`: already reconstructed); registers one
        trajectory per triangulated point on frame_idx.  Returns the world -> camera transform of img."""
        bootstrap = views if views is not None else self._bootstrap_views(baseline, img)
        T_rel = np.array(bootstrap.T)
        landmarks = np.array(bootstrap.point_cloud)
        if self.rescale_bootstrap and world_transform is not None:
            rel = np.asarray(world_transform, np.float64) @ hom_inv(np.asarray(baseline.pose, np.float64))
            scale = float(np.linalg.norm(rel[0:3, 3]))
            if scale > 0:
                T_rel[0:3, 3] *= scale
                landmarks[:, 0:3] *= scale
        T = T_rel @ baseline.pose
        num_pts = landmarks.shape[0]
        new_landmarks = (hom_inv(baseline.pose) @ landmarks.T).T
        store = self.keypoint_trajectories
        first = len(store.landmarks)
        store.landmarks.extend(new_landmarks[:, 0:3])
        store.create_trajectories(frame_idx, bootstrap.pts2[:, 0:2], T, np.arange(first, first + num_pts))
        self.inlier_count = num_pts
        return T

    def _process_frame(self, frame_idx: int, img: np.ndarray) -> None:
        store = self.keypoint_trajectories
        prev_img = self.frame_queue.get_head().img
        from_frame = store.latest_frame
        prev_keypoints, prev_trajectories, prev_landmarks = store.at_frame(from_frame)

        # the queue's images are never written to: the pyramid built for prev_img one step ago is kept
        T = None
        if self.fused_step:
            # tracker, filter and PnP in one library call; a frame with too few survivors (or no pose) falls
            # through to the forced bootstrap exactly as below
            tracked_pts, status, T, inliers = self.poseEstimator.track_and_pose(prev_img, img, prev_keypoints, prev_landmarks,
                                                                                reuse_pyramids=True, want_inliers=False)
        else:
            tracked_pts, status = PoseEstimation.KLT(prev_img, img, prev_keypoints, reuse_pyramids=True)
        if self.fused_step and T is not None:
            # the call's inlier mask is over the input rows already: no compaction of points and trajectory ids just to
            # index them again through the inlier list
            keep, n_tracked, rows = _tracked_frame_rows(status, self.poseEstimator.last_inlier_mask)
            tracked_landmarks = None        # prev_landmarks[keep], which only the depth statistic below would read
        else:
            rows = None
            tracked_pts, tracked_landmarks, trajectories = filter_tracked(status, tracked_pts, prev_landmarks,
                                                                          prev_trajectories)
            n_tracked = tracked_pts.shape[0]
        if n_tracked > 5:
            if not self.fused_step:
                T, inliers = self.poseEstimator.PnP(tracked_landmarks, tracked_pts)
            assert T is not None, "PNP RANSAC was not able to compute a pose from 2D - 3D correspondences"
            # the fused call reports the number of inliers; their list is not formed for the ratio alone
            ratio = inlier_ratio(inliers, self.inlier_count) if inliers is not None else \
                self.poseEstimator.last_inlier_count / self.inlier_count
        else:
            warnings.warn("too few points, forced bootstrap")
            prev_keyframe = self.keyframes[-1]
            T = self._bootstrap(prev_keyframe, frame_idx, img, world_transform=self.frame_queue.queue[-1].pose)
            self.bootstrap_idx.append(frame_idx)
            frame_state = FrameState(frame_idx, img, T, tracked_kps=n_tracked, is_key=True)
            self.frame_queue.add(frame_state)
            self.keyframes.append(frame_state)
            return

        if rows is not None:
            store.continue_rows(from_frame, rows, frame_idx, tracked_pts[rows], T)
        else:
            keep = np.unique(inliers.ravel())       # ascending: the order the reference visits them in
            store.tracked_to_many(trajectories[keep], frame_idx, tracked_pts[keep], T)

        is_key = False
        prev_keyframe = self.keyframes[-2] if len(self.keyframes) >= 2 else self.keyframes[-1]
        if tracked_landmarks is None:
            uncertainty = self._baseline_uncertainty(prev_keyframe.pose, T, prev_landmarks, keep)
        else:
            uncertainty = self._baseline_uncertainty(prev_keyframe.pose, T, tracked_landmarks)
        log.info("%d: tracked_pts: %5d, inlier_ratio: %.2f, baseline uncertainty: %.2f", frame_idx,
                 n_tracked, ratio, uncertainty)
        if uncertainty > params.MAX_BASELINE_UNCERTAINTY or ratio < params.MIN_INLIER_RATIO:
            is_key = True
            views = None
            if self.overlap_keyframe:
                prev_keyframe = self.frame_queue.get(2)
                features = None
                if self.bootstrap_correspondences == "sift" and _runtime.helper_thread:
                    # the bootstrap's two SIFT calls start on helper threads before the window is even assembled: they
                    # need the two images only, and each is one long library call (the interpreter lock is free
                    # meanwhile).  The rest of the two-view part stays on this thread: many short calls, which two
                    # threads would only hand the lock back and forth for (measured: the KLT route's loop a quarter
                    # slower with all of it on a helper).
                    features = BootstrapInitializer.sift_features_begin(prev_keyframe.img, img)
                # six trial steps up front instead of five: a window takes four to six, what is queued behind its end
                # is hidden by the bootstrap, a second round of steps would not be (eight cost their launches: 0.1 ms)
                try:
                    window = self._bundle_adjustment_begin(frame_idx, T, first_round=6)
                except Exception:
                    if features is not None:
                        features.discard()
                    raise
                try:
                    with _runtime.side_context():
                        views = self._bootstrap_views(prev_keyframe, img, features)
                finally:
                    self._bundle_adjustment_finish(window)
            else:
                self._bundle_adjustment(frame_idx, T)
                prev_keyframe = self.frame_queue.get(2)
            T = self._bootstrap(prev_keyframe, frame_idx, img,
                                world_transform=T if self.rescale_bootstrap else prev_keyframe.pose, views=views)
            self.bootstrap_idx.append(frame_idx)

        frame_state = FrameState(frame_idx, img, T, tracked_kps=n_tracked, is_key=is_key)
        self.frame_queue.add(frame_state)
        if is_key:
            self.keyframes.append(frame_state)

    def _baseline_uncertainty(self, T0: np.ndarray, T1: np.ndarray, landmarks: np.ndarray,
                              keep: Optional[np.ndarray] = None) -> float:
        return baseline_uncertainty(T0, T1, landmarks, keep)

    def _ba_window(self, frame_idx: int, T: np.ndarray):
        """Arrays of the BA window ending at frame_idx (continuousVO.py:273-313)."""
        store = self.keypoint_trajectories
        prev_keyframe = self.keyframes[max(-params.BUNDLE_ADJUSTMENT_KEYFRAME_LOOK_BACK, -len(self.keyframes))]
        look_back = frame_idx - prev_keyframe.idx
        if look_back >= self.frame_queue.size:
            look_back = self.frame_queue.size - 1
        poses, kps, trajs = [], [], []
        for i in range(look_back + 1):
            old_frame_idx = frame_idx - look_back + i
            pose = T if old_frame_idx == frame_idx else self.frame_queue.get(look_back - i - 1).pose
            poses.append(np.asarray(pose)[0:3, 0:4])
            kp, traj, _ = store.at_frame(old_frame_idx, with_landmarks=False)
            kps.append(kp)
            trajs.append(traj)
        # indices, concatenated keypoints and the landmark numbering in one pass of the library's host-side helper; ids too
        # sparse for its table (not what this pipeline produces) are numbered by window_indices
        workspace = self.__dict__.setdefault("_window_workspace", hip.HostWorkspace())
        done = hip.ba_window_assemble(kps, trajs, workspace)
        if done is not None:
            ci, pi, keypoints, traj_of_point = done
        else:
            ci, pi, traj_of_point = window_indices(trajs)
            keypoints = np.concatenate(kps).astype(np.float32).reshape(-1, 2)
        landmark_ids = store.traj2landmark.lookup(traj_of_point)
        # the window's arrays live in buffers that are used again by the next window (they have been handed to the solver
        # and uploaded by then): a window is megabytes, and fresh arrays of that size are page faults at every keyframe
        landmarks = store.landmarks.float32_rows(landmark_ids, out=workspace.take("window landmarks", landmark_ids.size, np.float32, 3))
        return look_back, np.asarray(poses, dtype=np.float32), landmarks, ci, pi, keypoints, landmark_ids

    def _bundle_adjustment(self, frame_idx: int, T: np.ndarray) -> np.ndarray:
        return self._bundle_adjustment_finish(self._bundle_adjustment_begin(frame_idx, T))

    def _bundle_adjustment_begin(self, frame_idx: int, T: np.ndarray, first_round: int = 0):
        """The window ending at frame_idx handed to the solver, its trial steps queued (first_round of them, 0: the
        solver's default); nothing is written back yet."""
        look_back, poses, landmarks, ci, pi, keypoints, landmark_ids = self._ba_window(frame_idx, T)
        begin = getattr(self.bundle_adjustment, "bundle_adjustment_begin", None)
        if begin is None:        # a solver with the reference's one call only
            return look_back, landmark_ids, None, self.bundle_adjustment.bundle_adjustment(poses, landmarks, ci, pi, keypoints)
        return look_back, landmark_ids, begin(poses, landmarks, ci, pi, keypoints, first_round=first_round), None

    def _bundle_adjustment_finish(self, window) -> np.ndarray:
        """Waits for the window and writes its poses and landmarks back (continuousVO.py:322-344)."""
        look_back, landmark_ids, in_flight, done = window
        adjusted_poses, adjusted_landmarks = in_flight.finish() if in_flight is not None else done
        hom = np.zeros((adjusted_poses.shape[0], 4, 4), adjusted_poses.dtype)
        hom[:, 0:3, :] = adjusted_poses
        hom[:, 3, 3] = 1
        for i in range(look_back):       # frames already in the queue; the current one is not yet
            self.frame_queue.get(look_back - i - 1).pose = hom[i].copy()
        self.keypoint_trajectories.landmarks.assign(landmark_ids, adjusted_landmarks)
        return hom[-1]

    @staticmethod
    def get_baseline_uncertainty(T: np.ndarray, point_cloud: np.ndarray) -> float:
        return float(np.linalg.norm(T[0:3, 3]) / np.mean(point_cloud[:, 2]))
