"""KeypointTrajectories -- the track store between the tracker and the pose / BA operators
(reference: vo_pipeline/keypointTrajectory.py:8-146, vo_pipeline/trajectory.py:4-37).

Same surface as the reference (landmarks, traj2landmark, on_frame, latest_frame,
create_trajectory, tracked_to, at_frame, get_active / get_all / get_inactive,
mean_trajectory_length), but array-backed: a frame's observations are (trajectory id, point)
arrays in registration order instead of a dict of per-keypoint objects, so at_frame() -- called
once per frame and once per window frame by the BA marshalling -- is three gathers instead of a
Python loop, and whole frames are registered with create_trajectories / tracked_to_many.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np


class _Growable:
    """Append-only (n, width) array with amortised growth."""

    def __init__(self, width: int, dtype):
        self._a = np.zeros((64, width) if width else (64,), dtype)
        self.n = 0

    def extend(self, rows: np.ndarray) -> None:
        k = rows.shape[0]
        if self.n + k > self._a.shape[0]:
            cap = max(2 * self._a.shape[0], self.n + k)
            grown = np.zeros((cap,) + self._a.shape[1:], self._a.dtype)
            grown[:self.n] = self._a[:self.n]
            self._a = grown
        self._a[self.n:self.n + k] = rows
        self.n += k

    @property
    def view(self) -> np.ndarray:
        return self._a[:self.n]


class LandmarkStore:
    """List-like store of 3-D landmarks: append, len, [i] read and write, assign(ids, xyz) for many at once; .array
    is a read-only (n,3) float64 view.  The tracker and the bundle-adjustment window read landmarks as float32 at every
    frame: a float32 mirror is kept for them (float32_rows), written by every path that writes the store -- which is
    why .array does not allow writing."""

    def __init__(self):
        self._g = _Growable(3, np.float64)
        self._m = _Growable(3, np.float32)

    def append(self, xyz) -> None:
        self.extend(np.asarray(xyz, np.float64).reshape(1, 3))

    def extend(self, xyz) -> None:
        xyz = np.asarray(xyz, np.float64).reshape(-1, 3)
        self._g.extend(xyz)
        self._m.extend(xyz)             # converted on assignment, as .astype(np.float32) would

    def assign(self, ids: np.ndarray, xyz) -> None:
        """landmarks[ids] = xyz for many ids at once (the write-back of a bundle-adjustment window)."""
        self._g.view[ids] = xyz
        self._m.view[ids] = xyz

    def float32_rows(self, ids: np.ndarray, out: np.ndarray = None) -> np.ndarray:
        """array[ids].astype(np.float32), from the mirror."""
        return np.take(self._m.view, ids, axis=0, out=out)

    def __len__(self) -> int:
        return self._g.n

    def __getitem__(self, i):
        return self.array[i]

    def __setitem__(self, i, xyz) -> None:
        self._g.view[i] = xyz
        self._m.view[i] = xyz

    @property
    def array(self) -> np.ndarray:
        view = self._g.view
        view.flags.writeable = False
        return view


class _TrajToLandmark:
    """dict-like view {trajectory id -> landmark id}; -1 in the backing array means unset."""

    def __init__(self):
        self._g = _Growable(0, np.int64)
        self.version = 0          # bumped when an existing trajectory's landmark changes (frames cache their lookups)

    def _grow_to(self, n: int) -> None:
        if n > self._g.n:
            self._g.extend(np.full(n - self._g.n, -1, np.int64))

    def __contains__(self, traj_idx) -> bool:
        t = int(traj_idx)
        return 0 <= t < self._g.n and self._g.view[t] >= 0

    def __getitem__(self, traj_idx) -> int:
        if traj_idx not in self:
            raise KeyError(traj_idx)
        return int(self._g.view[int(traj_idx)])

    def __setitem__(self, traj_idx, landmark_id) -> None:
        self._grow_to(int(traj_idx) + 1)
        self._g.view[int(traj_idx)] = int(landmark_id)
        self.version += 1

    def lookup(self, traj_ids: np.ndarray) -> np.ndarray:
        """Landmark id per trajectory id, -1 where unset."""
        return self._g.view[traj_ids] if traj_ids.size else np.zeros(0, np.int64)


class _FrameObservations:
    """(trajectory id, point) pairs of one frame in registration order; a trajectory registered
    twice on a frame keeps its first point (the reference's Trajectory.tracked_to appends nothing
    when the frame index does not advance)."""

    def __init__(self):
        self._ids: List[np.ndarray] = []
        self._pts: List[np.ndarray] = []
        self.served = None        # what at_frame last returned for this frame, with the landmark-map version it saw

    def add(self, ids: np.ndarray, pts: np.ndarray) -> None:
        self._ids.append(ids)
        self._pts.append(pts)
        self.served = None

    def _consolidate(self) -> None:
        if len(self._ids) > 1:
            ids, pts = np.concatenate(self._ids), np.concatenate(self._pts)
            _, first = np.unique(ids, return_index=True)
            if first.size != ids.size:
                first.sort()
                ids, pts = ids[first], pts[first]
            self._ids, self._pts = [ids], [pts]

    @property
    def ids(self) -> np.ndarray:
        self._consolidate()
        return self._ids[0] if self._ids else np.zeros(0, np.int64)

    @property
    def pts(self) -> np.ndarray:
        self._consolidate()
        return self._pts[0] if self._pts else np.zeros((0, 2), np.float64)

    def keys(self):
        return self.ids.tolist()

    def __len__(self) -> int:
        return self.ids.shape[0]

    def __contains__(self, traj_idx) -> bool:
        return bool(np.any(self.ids == int(traj_idx)))


class Trajectory:
    """Record of one trajectory, assembled on demand (vo_pipeline/trajectory.py:4-37)."""

    def __init__(self, traj_idx, init_idx, final_idx, pts, init_transform, final_transform):
        self.traj_idx, self.init_idx, self.final_idx = traj_idx, init_idx, final_idx
        self.pts = pts
        self.init_point, self.final_point = pts[0], pts[-1]
        self.init_transform, self.final_transform = init_transform, final_transform


class KeypointTrajectories:
    def __init__(self):
        self.landmarks = LandmarkStore()
        self.traj2landmark = _TrajToLandmark()
        self.on_frame: Dict[int, _FrameObservations] = {}
        self.latest_frame = 0
        self._next_trajectory_idx = 0
        self._span = _Growable(2, np.int64)        # (init frame, final frame) per trajectory
        self._tf = _Growable(2, np.int64)          # (init transform id, final transform id)
        self._transforms: List[np.ndarray] = []

    # ---- registration ------------------------------------------------------------
    def _transform_id(self, transform) -> int:
        self._transforms.append(np.float32(transform))
        return len(self._transforms) - 1

    def _frame(self, frame_idx: int) -> _FrameObservations:
        obs = self.on_frame.get(frame_idx)
        if obs is None:
            obs = self.on_frame[frame_idx] = _FrameObservations()
        return obs

    def create_trajectories(self, frame_idx: int, pts: np.ndarray, transform: np.ndarray,
                            landmark_ids: Optional[np.ndarray] = None) -> np.ndarray:
        """Start one trajectory per row of pts (n,2) on frame_idx -> their ids (n,)."""
        pts = np.asarray(pts).reshape(-1, 2)
        n = pts.shape[0]
        ids = np.arange(self._next_trajectory_idx, self._next_trajectory_idx + n, dtype=np.int64)
        self._next_trajectory_idx += n
        tf = self._transform_id(transform)
        self._span.extend(np.full((n, 2), frame_idx, np.int64))
        self._tf.extend(np.full((n, 2), tf, np.int64))
        self.traj2landmark._grow_to(self._next_trajectory_idx)
        if landmark_ids is not None:
            self.traj2landmark._g.view[ids] = np.asarray(landmark_ids, np.int64)
        if n:
            self._frame(frame_idx).add(ids, pts.copy())
        self._next_frame(frame_idx)
        return ids

    def tracked_to_many(self, traj_ids: np.ndarray, frame_idx: int, pts: np.ndarray, transform: np.ndarray) -> None:
        """Extend the trajectories traj_ids (n,) to frame_idx with the points pts (n,2)."""
        ids = np.asarray(traj_ids, np.int64).ravel()
        pts = np.asarray(pts).reshape(-1, 2)
        assert ids.shape[0] == pts.shape[0]
        if ids.size:
            assert ids.min() >= 0 and ids.max() < self._next_trajectory_idx, "unknown trajectory"
            self._span.view[ids, 1] = frame_idx
            self._tf.view[ids, 1] = self._transform_id(transform)
            self._frame(frame_idx).add(ids, pts.copy())
        self._next_frame(frame_idx)

    def continue_rows(self, from_frame: int, rows: np.ndarray, frame_idx: int, pts: np.ndarray, transform: np.ndarray) -> None:
        """tracked_to_many for the trajectories at rows `rows` of what at_frame(from_frame) last returned (not in the
        reference): their ids and landmark links are known from that answer, so what at_frame(frame_idx) will be asked
        for is set up here instead of being looked up again -- the tracked frame's path through the store."""
        served = self.on_frame[from_frame].served if from_frame in self.on_frame else None
        fresh = frame_idx not in self.on_frame or len(self.on_frame[frame_idx]._ids) == 0
        if served is None or served[0] != self.traj2landmark.version or not fresh:
            ids = self.at_frame(from_frame, with_landmarks=False)[1][rows]
            return self.tracked_to_many(ids, frame_idx, pts, transform)
        _, _, ids32, lid = served
        ids32, lid = ids32[rows], lid[rows]
        pts = np.array(pts, np.float32).reshape(-1, 2)          # the store's own copy
        assert ids32.shape[0] == pts.shape[0]
        if ids32.size:
            ids = ids32.astype(np.int64)
            self._span.view[ids, 1] = frame_idx
            self._tf.view[ids, 1] = self._transform_id(transform)
            obs = self._frame(frame_idx)
            obs.add(ids, pts)
            obs.served = (self.traj2landmark.version, pts, ids32, lid)
        self._next_frame(frame_idx)

    def create_trajectory(self, frame_idx: int, pt: np.ndarray, transform: np.ndarray,
                          landmark_id: Optional[int] = None) -> Tuple[Trajectory, int]:
        lid = None if landmark_id is None else np.array([landmark_id])
        traj_idx = int(self.create_trajectories(frame_idx, np.asarray(pt).reshape(1, 2), transform, lid)[0])
        return self.trajectory(traj_idx), traj_idx

    def tracked_to(self, traj_idx: int, frame_idx: int, pt: np.ndarray, transform: np.ndarray,
                   landmark_id=None) -> Trajectory:
        if not 0 <= int(traj_idx) < self._next_trajectory_idx:
            raise KeyError(traj_idx)
        self.tracked_to_many(np.array([traj_idx]), frame_idx, np.asarray(pt).reshape(1, 2), transform)
        if landmark_id is not None:
            self.traj2landmark[traj_idx] = landmark_id
        return self.trajectory(traj_idx)

    def _next_frame(self, frame_idx: int) -> None:
        self.latest_frame = max(self.latest_frame, frame_idx)

    # ---- queries -----------------------------------------------------------------
    def at_frame(self, frame_idx: int, with_landmarks: bool = True) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
        """Keypoints with known landmarks on frame_idx ->
        (keypoints (N,2) float32, trajectory ids (N,) int32, landmarks (N,3) float32).
        with_landmarks=False (not in the reference) skips the landmark gather and returns None in its place,
        for callers that only want the observations (the bundle-adjustment window)."""
        assert frame_idx in self.on_frame, f"frame_idx {frame_idx} needs to have keypoints associated with it"
        obs = self.on_frame[frame_idx]
        # a frame is asked for once when it is tracked from and again by every bundle-adjustment window that reaches
        # back to it: the converted arrays are kept until the frame gets observations or a landmark link changes
        # (callers treat them as read-only)
        if obs.served is not None and obs.served[0] == self.traj2landmark.version:
            _, pts, kept, lid = obs.served
        else:
            ids = obs.ids
            lid = self.traj2landmark.lookup(ids)
            keep = lid >= 0
            if keep.all():
                pts, kept = obs.pts.astype(np.float32, copy=False), ids.astype(np.int32)
            else:
                pts, kept, lid = obs.pts[keep].astype(np.float32), ids[keep].astype(np.int32), lid[keep]
            obs.served = (self.traj2landmark.version, pts, kept, lid)
        if not with_landmarks:
            return pts, kept, None
        return pts, kept, self.landmarks.float32_rows(lid).reshape(-1, 3)

    def trajectory(self, traj_idx: int) -> Trajectory:
        """Per-trajectory record; the point of a skipped frame repeats the next registered one."""
        t = int(traj_idx)
        init, final = (int(v) for v in self._span.view[t])
        pts, pending = [], 0
        for f in range(init, final + 1):
            obs = self.on_frame.get(f)
            hit = np.nonzero(obs.ids == t)[0] if obs is not None else ()
            if len(hit):
                pts += [obs.pts[hit[0]]] * (pending + 1)
                pending = 0
            else:
                pending += 1
        a, b = self._tf.view[t]
        return Trajectory(t, init, final, pts, self._transforms[a], self._transforms[b])

    @property
    def trajectories(self) -> Dict[int, Trajectory]:
        return {t: self.trajectory(t) for t in range(self._next_trajectory_idx)}

    def _latest(self) -> Tuple[np.ndarray, np.ndarray]:
        obs = self.on_frame.get(self.latest_frame)
        ids = obs.ids if obs is not None else np.zeros(0, np.int64)
        return ids, self.traj2landmark.lookup(ids)

    def get_active(self) -> List[np.ndarray]:
        _, lid = self._latest()
        return list(self.landmarks._g.view[lid[lid >= 0]])

    def get_all(self) -> List[np.ndarray]:
        ids, lid = self._latest()
        if np.any(lid < 0):
            raise KeyError(int(ids[lid < 0][0]))
        return list(self.landmarks._g.view[lid])

    def get_inactive(self) -> List[np.ndarray]:
        return []   # trajectories without a landmark have none to return

    def mean_trajectory_length(self) -> float:
        ids, lid = self._latest()
        span = self._span.view[ids[lid >= 0]]
        return float(np.mean(span[:, 1] - span[:, 0]))
