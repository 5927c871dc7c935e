"""FrameQueue -- bounded history of FrameState objects, newest last; get(0) is the newest
(reference: utils/frameQueue.py:6-41)."""
from collections import deque


class FrameQueue:
    def __init__(self, size: int):
        assert size > 0
        self.size = size
        self.queue = deque(maxlen=size)
        self._i = -1

    @property
    def length(self) -> int:
        return len(self.queue)

    def add(self, frame_state) -> None:
        self.queue.append(frame_state)      # the oldest entry falls out when the queue is full

    def get(self, idx: int):
        """idx frames back from the newest one."""
        assert idx >= 0
        return self.queue[len(self.queue) - idx - 1]

    def get_head(self):
        return self.get(0)

    def __iter__(self):
        return self

    def __next__(self):
        self._i += 1
        if self._i < len(self.queue):
            return self.queue[self._i]
        self._i = -1
        raise StopIteration()
