"""FrameState -- per-frame record of the pipeline (reference: vo_pipeline/frameState.py:5-25)."""
from typing import Optional

import numpy as np


class FrameState:

    def __init__(self, idx: int, img: np.ndarray, pose: np.ndarray, tracked_kps: int = 0,
                 keypoints: Optional[np.ndarray] = None, descriptors: Optional[np.ndarray] = None, is_key=False):
        assert idx >= 0, "Frame index must be non-negative"
        assert pose.shape == (4, 4), "pose has to be a homogeneous (4, 4) tranfrom matrix"
        self.idx = idx
        self.img = img
        self.pose = pose            # world -> camera
        self.keypoints = keypoints
        self.descriptors = descriptors
        self.is_key = is_key
        self.num_tracked_kps = tracked_kps
