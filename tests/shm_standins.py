"""Stand-ins for the reference's shared-memory frame objects (`watsor/stream/share.py:27-90`) built from the same
`multiprocessing.sharedctypes` primitives, for the GPU box where no Watsor is installed: `Frame` = `Value(Header)` +
`Array('B', w*h*3)` + a latch that counts its `next()` calls in shared memory; `FrameBuffer` = a list of them.
Only what the detector worker touches is here (`header`, `image`, `latch.next()`, `get_numpy_image`, `frames`)."""
from collections import namedtuple
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


def spawn_context():
    return get_context("spawn")        # the reference's start method: watsor/main.py:474
