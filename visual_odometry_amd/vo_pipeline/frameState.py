"""FrameState -- per-frame record of the pipeline (reference: vo_pipeline/frameState.py:5-25)."""
from typing import Optional

import numpy as np


class FrameState:
    """idx: frame number; img: the grayscale frame; pose: world -> camera (4,4); is_key: keyframe flag;
    num_tracked_kps: keypoints tracked into this frame (constructor argument `tracked_kps`, as in the
    reference); keypoints / descriptors: optional per-frame features."""

    def __init__(self, idx: int, img: np.ndarray, pose: np.ndarray, tracked_kps: int = 0,
                 keypoints: Optional[np.ndarray] = None, descriptors: Optional[np.ndarray] = None, is_key=False):
        if idx < 0:
            raise AssertionError(f"FrameState: frame index {idx} is negative")
        if np.shape(pose) != (4, 4):
            raise AssertionError(f"FrameState: pose must be a homogeneous (4,4) transform, got shape {np.shape(pose)}")
        self.idx, self.img, self.pose = idx, img, pose
        self.keypoints, self.descriptors = keypoints, descriptors
        self.is_key = is_key
        self.num_tracked_kps = tracked_kps
