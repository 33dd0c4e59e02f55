"""Tracker -> fusion wire: the reference's `DataTrans<T>` singleton queue
(src/DataTrans.h:10-83) -- bounded at 30, `product` drops the OLDEST element when full,
`consumption` blocks until an element arrives -- and the TestSystem feed loop
(Map2DFusion/Map2DFusion.cpp:309-327: feed only while queueSize() < 2, paced at Video.fps)."""
import collections
import threading
import time


class DataTrans:
    def __init__(self, max_size=30):                       # DataTrans.h:36
        self.max_size = max_size
        self._q = collections.deque()
        self._cv = threading.Condition()
        self.dropped = 0

    def product(self, v):                                  # DataTrans.h:54-68
        with self._cv:
            while len(self._q) >= self.max_size:
                self._q.popleft()
                self.dropped += 1
            self._q.append(v)
            self._cv.notify()

    def consumption(self, timeout=None):                   # DataTrans.h:70-83 (blocking)
        with self._cv:
            if not self._cv.wait_for(lambda: len(self._q) > 0, timeout):
                return None
            return self._q.popleft()

    def size(self):
        with self._cv:
            return len(self._q)


def feed_loop(map2d, source, fps=0.0, stop=lambda: False):
    """TestSystem's auto-feed loop: `source()` returns (image, pose) or None at the end."""
    period = 1.0 / fps if fps > 0 else 0.0
    fed = 0
    while not stop():
        t0 = time.perf_counter()
        if map2d.queueSize() < 2:                          # Map2DFusion.cpp:313
            item = source()
            if item is None:
                break
            map2d.feed(item[0], item[1])
            fed += 1
        if period:
            time.sleep(max(0.0, period - (time.perf_counter() - t0)))
    return fed
