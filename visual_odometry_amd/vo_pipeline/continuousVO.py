"""Marshalling between the track store and the hot-path operators -- the parts of
the reference's ContinuousVO (vo_pipeline/continuousVO.py:157-344) that shape the
arrays the kernels consume.  The orchestrator state machine itself is outside the
accelerated path."""
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ..utils.matrix import hom_inv


def filter_tracked(status: np.ndarray, tracked_pts: np.ndarray, *per_point: np.ndarray):
    """Keep rows with status == 1 in their original order (continuousVO.py:168-172)."""
    keep = np.asarray(status).ravel() == 1
    return (tracked_pts[keep],) + tuple(a[keep] for a in per_point)


def inlier_ratio(inliers: np.ndarray, inlier_count: int) -> float:
    """len(inliers) / inlier_count with the count fixed at the last bootstrap (continuousVO.py:178)."""
    return inliers.shape[0] / inlier_count


def baseline_uncertainty(T0: np.ndarray, T1: np.ndarray, landmarks: np.ndarray) -> float:
    """|camera-centre displacement| / mean positive landmark depth in camera 0 (continuousVO.py:249-271);
    inf when no landmark has positive depth."""
    T0_inv, T1_inv = hom_inv(T0), hom_inv(T1)
    normal = T0[0:3, 0:3].T @ np.array([0.0, 0.0, 1.0])
    centred = landmarks - T0_inv[0:3, 3]
    d = centred @ normal
    d = d[d > 0]
    if d.size == 0:
        return np.inf
    return float(np.linalg.norm(T1_inv[:3, 3] - T0_inv[:3, 3]) / np.mean(d))


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
