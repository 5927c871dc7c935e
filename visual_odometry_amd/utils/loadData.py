"""Dataset container of the pipeline (reference: utils/loadData.py:17-22) and a synthetic
source for it.  The reference's KITTI / Malaga / parking loaders decode image files with cv2
and are not part of this build; any iterator of (K, gray uint8 image) serves."""
import enum
from typing import Iterator, Optional, Tuple

import numpy as np


class DatasetType(enum.Enum):
    KITTI = 0,
    MALAGA = 1,
    PARKING = 2


class Dataset:
    def __init__(self, K: np.ndarray, frames: Iterator[Tuple[np.ndarray, np.ndarray]], T: Optional[np.ndarray]):
        assert K.shape == (3, 3), "K has to be a 3x3 matrix"
        self.K = K
        self.frames = frames
        self.T = T      # ground-truth camera-to-world transforms (N,4,4) or None


def synthetic_parking(n_frames: int = 40, seed: int = 5, **kw) -> Dataset:
    """640x480 ray-cast box scene with the parking camera matrix and its ground-truth poses."""
    from .. import synth
    scene = synth.BoxScene(n_frames=n_frames, seed=seed, **kw)
    return Dataset(np.asarray(scene.K, np.float32), scene.frames(), np.asarray(scene.poses, np.float32))
