"""FrameQueue -- bounded history of FrameState objects, newest last; get(0) is the newest
(reference: utils/frameQueue.py:6-41)."""
from collections import deque
from typing import Iterator


class FrameQueue:
    """Fixed-capacity history on a deque: appending to a full queue drops the oldest frame.
    `queue` is oldest-first, like the reference's list; iteration walks it oldest to newest."""

    def __init__(self, size: int):
        if size <= 0:
            raise AssertionError(f"FrameQueue needs a positive capacity, got {size}")
        self.size = size
        self.queue = deque(maxlen=size)

    @property
    def length(self) -> int:
        return len(self.queue)

    def __len__(self) -> int:
        return len(self.queue)

    def add(self, frame_state) -> None:
        self.queue.append(frame_state)

    def get(self, idx: int):
        """The frame `idx` steps back from the newest (0 = newest); IndexError beyond the history."""
        if idx < 0:
            raise AssertionError(f"FrameQueue.get: negative look-back {idx}")
        if idx >= len(self.queue):
            raise IndexError(f"FrameQueue.get: look-back {idx} with {len(self.queue)} frames queued")
        return self.queue[-1 - idx]

    def get_head(self):
        return self.get(0)

    def __iter__(self) -> Iterator:
        return iter(tuple(self.queue))      # a snapshot: adding frames while iterating is allowed
