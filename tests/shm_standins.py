"""Stand-ins for the reference's shared-memory frame objects (`watsor/stream/share.py:27-90`) built from the same
`multiprocessing.sharedctypes` primitives, for the GPU box where no Watsor is installed: `Frame` = `Value(Header)` +
`Array('B', w*h*3)` + a latch that counts its `next()` calls in shared memory; `FrameBuffer` = a list of them.
Only what the detector worker touches is here (`header`, `image`, `latch.next()`, `get_numpy_image`, `frames`)."""
from collections import namedtuple
from ctypes import Structure, c_double
from multiprocessing import get_context

import numpy as np

from watsor_amd.share import Header

Payload = namedtuple("Payload", ["sender", "frame_index"])     # watsor/stream/work.py:42


class Latch:
    def __init__(self, ctx):
        self.steps = ctx.Value("i", 0)

    def next(self, *a):
        with self.steps.get_lock():
            self.steps.value += 1


class Frame:
    def __init__(self, ctx, width, height, channels=3):
        self.header = ctx.Value(Header, width, height, channels, 0)
        self.image = ctx.Array("B", width * height * channels)
        self.latch = Latch(ctx)

    def get_numpy_image(self, dtype=None):                      # share.py:68-73
        image_shape = (self.header.height, self.header.width, self.header.channels)
        return image_shape, np.frombuffer(self.image.get_obj(), dtype).reshape(image_shape)


class FrameBuffer:
    def __init__(self, ctx, maxsize, width, height, channels=3):                # share.py:76-81
        self.frames = [Frame(ctx, width, height, channels) for _ in range(maxsize)]


class Gauge:
    """FramesPerSecond / InferenceTime stand-in (`share.py:196-238`): remembers what it was called with."""
    def __init__(self, ctx):
        self.count = ctx.Value("i", 0)
        self.total = ctx.Value("d", 0.0)

    def __call__(self, value=None):
        with self.count.get_lock():
            self.count.value += 1
            if isinstance(value, (int, float)) and not isinstance(value, bool):
                self.total.value += float(value)


# ---- stand-ins that COST what the reference's objects cost (bench.py's `worker_spawned` leg): the same multiprocessing
# primitives taken in the same order, so that frames/s through the worker loop on the GPU box mean what they would under an
# installed Watsor.  tests/test_reference_plumbing.py holds their per-call time against the reference's own classes.
class StateLatchStandIn:
    """`StateLatch.next()` (watsor/stream/sync.py:62-104): a Condition over the frame's RLock, the state and two counters as
    Values under that lock, an inner zero-time wait on the count-down condition, a notify_all -- plus the step count the tests read
    and (bench) a stamp of when the frame came back, against the `epoch` its producer wrote into the header."""

    def __init__(self, ctx, lock, header=None, samples=None):
        self._cond = ctx.Condition(lock)
        self._state = ctx.Value("i", 1, lock=lock)
        self._count = ctx.Value("i", 0, lock=lock)
        self._count_cond = ctx.Condition(lock)
        self._count_max = ctx.Value("i", 0, lock=lock)
        self.steps = ctx.Value("i", 0, lock=lock)
        self._header, self._samples = header, samples

    def next(self, count_down=0):
        self._cond.acquire()
        try:
            self._count_max.value = max(self._count_max.value, count_down)
            self._count_cond.acquire()
            reached = self._count_cond.wait_for(lambda: self._count.value <= 0, 0)
            self._count_cond.release()
            if reached:
                self._state.value = self._state.value % 3 + 1
                self._count_cond.acquire()
                self._count.value = self._count_max.value
                self._count_cond.release()
                self._count_max.value = 0
            self.steps.value += 1
            if self._samples is not None and (self.steps.value & 7) == 0:       # every 8th step: enqueue -> latch, seconds
                from time import time
                k = self._samples[0]
                self._samples[1 + int(k) % (len(self._samples) - 1)] = time() - self._header.epoch
                self._samples[0] = k + 1
            self._cond.notify_all()
        finally:
            self._cond.release()


class CostlyFrame(Frame):
    """`Frame` with the reference's locking: header, image and latch share one RLock (share.py:35-41)."""

    def __init__(self, ctx, width, height, channels=3, samples=None):
        lock = ctx.RLock()
        self.header = ctx.Value(Header, width, height, channels, 0, lock=lock)
        self.image = ctx.Array("B", width * height * channels, lock=lock)
        self.latch = StateLatchStandIn(ctx, lock, self.header, samples)


class CostlyFrameBuffer:
    def __init__(self, ctx, maxsize, width, height, channels=3, samples=None):
        self.frames = [CostlyFrame(ctx, width, height, channels, samples) for _ in range(maxsize)]


class Cell(Structure):
    _fields_ = [("time", c_double), ("value", c_double)]


class CostlyGauge:
    """`FramesPerSecond` / `InferenceTime` (share.py:161-238): a ring of 100 (time, value) cells in shared memory under one
    RLock; every call stamps a cell, expires old ones and computes the figure -- for `mean=True` (InferenceTime) with a Python
    loop over all 100 cells.  `count` / `total` as in `Gauge`."""

    def __init__(self, ctx, mean=False, maxlen=100, timeframe=10.0):
        self._lock = ctx.RLock()
        self._cells = ctx.Array(Cell, [(0.0, 0.0)] * maxlen, lock=self._lock)
        self._index = ctx.Value("i", 0, lock=self._lock)
        self._start = ctx.Value("i", 0, lock=self._lock)
        self._length = ctx.Value("i", 0, lock=self._lock)
        self.count = ctx.Value("i", 0, lock=self._lock)
        self.total = ctx.Value("d", 0.0, lock=self._lock)
        self._maxlen, self._timeframe, self._mean = maxlen, timeframe, mean

    def _inc(self, v):
        v.value = v.value + 1
        if v.value >= self._maxlen:
            v.value = 0

    def __call__(self, value=None):
        from time import time
        self._lock.acquire()
        try:
            now = time()
            if value is not None:
                self._cells[self._index.value] = (now, value)
                self._inc(self._index)
                if self._length.value < self._maxlen:
                    self._length.value = self._length.value + 1
                if self._length.value == self._maxlen:
                    self._inc(self._start)
                self.count.value += 1
                if isinstance(value, (int, float)) and not isinstance(value, bool):
                    self.total.value += float(value)
            while self._length.value > 0 and self._cells[self._start.value].time + self._timeframe < now:
                self._cells[self._start.value] = (0, 0)
                if self._length.value < self._maxlen:
                    self._inc(self._start)
                self._length.value = self._length.value - 1
            if self._length.value == 0:
                return 0.0
            if self._mean:
                acc = 0.0
                for i in range(self._maxlen):
                    acc += self._cells[i].value
                return acc / self._length.value
            dt = self._cells[self._index.value - 1].time - self._cells[self._start.value].time
            return self._length.value / dt if dt else 0.0
        finally:
            self._lock.release()


class BalancedQueueStandIn:
    """`BalancedQueue` (watsor/stream/sync.py:144-166): one real queue, one semaphore per sender -- a camera has at most one
    frame queued, `get` hands the slot back."""

    def __init__(self, delegate, semaphores, sender=None):
        self._q, self._sems, self._sender = delegate, semaphores, sender

    def put(self, obj, block=True, timeout=None):
        from queue import Full
        if not self._sems[self._sender].acquire(block, timeout):
            raise Full
        self._q.put((self._sender, obj), block, timeout)

    def get(self, block=True, timeout=None):
        sender, obj = self._q.get(block, timeout)
        self._sems[sender].release()
        return obj

    def get_nowait(self):
        return self.get(False)


def spawn_context():
    return get_context("spawn")        # the reference's start method: watsor/main.py:474


def affinity_echo(kwargs, out):
    """Child-process half of tests/test_worker_logic.py::test_camera_affinity_travels_into_spawned_workers: what a spawned detector process
    receives in `kwargs['hip_affinity']` -- the owner table and working queues."""
    aff = kwargs["hip_affinity"]
    got = aff["side"][aff["index"]].get(timeout=20)
    aff["side"][1 - aff["index"]].put(("forwarded-by-%d" % aff["index"], got))
    out.put((aff["index"], aff["count"], sorted(aff["owners"].items())))
