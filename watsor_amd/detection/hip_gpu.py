"""`HipObjectDetector` -- the MI355X detector plugin.

Satisfies the duck-typed protocol every reference detector implements
(`watsor/detection/tensorflow_cpu.py:13,64-92`, `tensorrt_gpu.py:20,55-91`,
`tensorflow_lite_cpu.py:16,25-62`, `edge_tpu.py:17,24-57`):

    __init__(model_path, device)    raises FileNotFoundError when model/mi355x.bin is absent
    device_name -> str
    __enter__ / __exit__            frees device state
    detect(image_shape, image_np, detections) -> milliseconds

so `ObjectDetector._run/_next_frame` (`watsor/detection/detector.py:84-112`) drives it unchanged.
`detect()` hands the frame pointer and the ctypes `Detection[100]` array straight to
`wz_detect_batch`; the 100 rows (class, score, pixel box -- `tensorflow_cpu.py:79-90`) are produced
on the GPU and written in place.  `detect_batch()` is the same call for several frames (BASELINE
config 2: batch = 8), used by `BatchedObjectDetector`.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np

from ..runtime import HipEngine
from ..share import Detection

ENGINE_FILE = "mi355x.bin"       # the analogue of gpu.trt (watsor/detection/detector.py:44)


class HipObjectDetector:
    """Performs object detection on AMD Instinct MI355X GPUs (hand-written HIP kernels)."""

    def __init__(self, model_path, device: int = 0, max_batch: Optional[int] = None,
                 max_width: Optional[int] = None, max_height: Optional[int] = None):
        engine_path = os.path.join(model_path, ENGINE_FILE)
        if not os.path.isfile(engine_path):
            raise FileNotFoundError(engine_path)
        max_batch = max_batch or int(os.environ.get("WATSOR_HIP_MAX_BATCH", "8"))
        max_width = max_width or int(os.environ.get("WATSOR_HIP_MAX_WIDTH", "1920"))
        max_height = max_height or int(os.environ.get("WATSOR_HIP_MAX_HEIGHT", "1080"))
        self.__engine = HipEngine(engine_path, device, max_batch, max_width, max_height)
        self.__device = device

    @property
    def engine(self) -> HipEngine:
        return self.__engine

    @property
    def max_batch(self) -> int:
        return self.__engine.max_batch

    @property
    def device_name(self):
        return self.__engine.device_name

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.__engine.close()

    def detect(self, image_shape, image_np, detections: List[Detection]):
        return self.__engine.detect_batch([image_np.reshape(image_shape)], [detections])

    def detect_batch(self, image_shapes: Sequence, images: Sequence[np.ndarray], detections: Sequence,
                     cameras: Optional[Sequence[int]] = None, passes: Optional[Sequence[np.ndarray]] = None):
        frames = [im.reshape(sh) for sh, im in zip(image_shapes, images)]
        return self.__engine.detect_batch(frames, detections, cameras, passes)
