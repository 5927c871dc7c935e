"""PoseGraphOptimizer -- the reference's vo_pipeline/poseOptimization.py:5-10 is an
empty stub (`__init__(frame_queue)` and `optimize()` both `pass`).  The class is
kept importable with that constructor; optimize() performs motion-only
refinement: every queued frame's pose is refined against its tracked landmarks
(landmarks fixed) with the 6x6 normal equations of the GPU BA kernel."""
from typing import Optional

import numpy as np

from . import _runtime, ransac


class PoseGraphOptimizer:
    def __init__(self, frame_queue, K: Optional[np.ndarray] = None, keypoint_trajectories=None):
        self.frame_queue = frame_queue
        self.K = K
        self.keypoint_trajectories = keypoint_trajectories

    def optimize(self, iters: int = 10) -> int:
        """Refine the pose of each frame in the queue in place; returns the number of frames refined.
        Needs K and a track store with at_frame(idx) -> (keypoints (N,2), traj ids, landmarks (N,3))."""
        if self.frame_queue is None or self.K is None or self.keypoint_trajectories is None:
            return 0
        ctx = _runtime.context()
        frames = getattr(self.frame_queue, "queue", self.frame_queue)
        done = 0
        for fs in frames:
            try:
                kp, _, lm = self.keypoint_trajectories.at_frame(fs.idx)
            except (AssertionError, KeyError):
                continue
            if kp.shape[0] < 6:
                continue
            T = np.asarray(fs.pose, np.float64)
            R, t = ransac.refine_pose(ctx, np.asarray(lm, np.float32), np.asarray(kp, np.float32),
                                      np.asarray(self.K, np.float64), T[:3, :3], T[:3, 3], iters)
            pose = np.eye(4)
            pose[:3, :3], pose[:3, 3] = R, t
            fs.pose = pose
            done += 1
        return done
