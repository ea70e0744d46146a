"""Detector factory + worker for the MI355X family: a superset of `watsor/detection/detector.py`.

`create_object_detectors` keeps the reference's signature and file-gate convention
(`detector.py:12-55`: `edgetpu.tflite` -> Coral, `gpu.trt` -> CUDA, else CPU) and adds one more
gate in front: `mi355x.bin` in the model path -> one detector process per AMD GPU.  When that file
is absent (or no AMD GPU / HIP library is present) it defers to the reference factory unchanged.

`BatchedObjectDetector` is the reference's `ObjectDetector` worker (`detector.py:58-112`) with one
difference: after the blocking `get(timeout=1)` it drains up to `max_batch - 1` more payloads with
`get_nowait()` and runs them through ONE `detect_batch()` call (BASELINE config 2: batch = 8
frames on one GPU), then performs exactly one `frame.latch.next()` per dequeued payload
(`detector.py:111-112`), in `finally`, like the reference.

This module needs the reference package (`watsor.stream`) at run time -- it is a drop-in for an
installed Watsor, see INTEGRATION.md.  Everything below `detect_batch()` does not.
"""
from __future__ import annotations

from os import path
from queue import Empty

from numpy import uint8

from watsor_amd.detection.devices import hip_gpus
from watsor_amd.detection.hip_gpu import ENGINE_FILE

try:  # the reference runtime; absent on a bare GPU box
    from watsor.detection import detector as _ref
    from watsor.stream.work import Work
    _HAVE_WATSOR = True
except ImportError:  # pragma: no cover - exercised on the GPU box
    _ref = None
    _HAVE_WATSOR = False


def _require_watsor():
    if not _HAVE_WATSOR:
        raise ImportError("watsor_amd.detection.detector plugs into an installed Watsor "
                          "(watsor.detection.detector / watsor.stream); it was not found on sys.path")


if _HAVE_WATSOR:

    class BatchedObjectDetector(_ref.ObjectDetector):
        """`ObjectDetector` that dequeues several frames and runs one batched detection."""

        def _process(self, frame_queue, stop_event, frame_buffers, fps, inference_time, object_detector,
                     *args, **kwargs):
            try:
                first = frame_queue.get(timeout=1)
            except Empty:
                return self._no_frame(stop_event, frame_buffers, fps, inference_time, object_detector, *args, **kwargs)
            if first is None:
                return
            payloads = [first]
            limit = getattr(object_detector, "max_batch", 1)
            if hasattr(object_detector, "detect_batch"):
                while len(payloads) < limit:
                    try:
                        nxt = frame_queue.get_nowait()
                    except Empty:
                        break
                    if nxt is not None:
                        payloads.append(nxt)
            return self._next_frames(payloads, stop_event, frame_buffers, fps, inference_time, object_detector)

        def _next_frames(self, payloads, stop_event, frame_buffers, fps, inference_time, object_detector):
            frames = [frame_buffers[p.sender].frames[p.frame_index] for p in payloads]
            try:
                if len(frames) == 1 or not hasattr(object_detector, "detect_batch"):
                    for frame in frames:
                        image_shape, image_np = frame.get_numpy_image(uint8)
                        time_of_inference = object_detector.detect(image_shape, image_np, frame.header.detections)
                        inference_time(value=time_of_inference)
                        fps(value=True)
                else:
                    shapes, images, rows = [], [], []
                    for frame in frames:
                        image_shape, image_np = frame.get_numpy_image(uint8)
                        shapes.append(image_shape)
                        images.append(image_np)
                        rows.append(frame.header.detections)
                    time_of_inference = object_detector.detect_batch(shapes, images, rows)
                    for _ in frames:
                        inference_time(value=time_of_inference)
                        fps(value=True)
            finally:
                for frame in frames:
                    frame.latch.next()


def create_object_detectors(delegate_class, stop_event, log_queue, frame_queue, frame_buffers, model_path,
                            kwargs=None):
    """Creates all available detectors: AMD GPUs first when `mi355x.bin` is provided, otherwise
    whatever the reference factory finds (Coral, CUDA, CPU).  Same arguments and return value as
    `watsor.detection.detector.create_object_detectors` (detector.py:12-55)."""
    _require_watsor()
    detectors = []
    if kwargs is None:
        kwargs = {}

    if path.isfile(path.join(model_path, ENGINE_FILE)):
        for device, clazz in hip_gpus():
            detectors.append(BatchedObjectDetector(
                delegate_class, "detector{}".format(len(detectors) + 1), stop_event, log_queue, frame_queue,
                frame_buffers, kwargs={**kwargs, 'detector_class': clazz, 'detector_args': (model_path, device)}))

    if _ref._ALWAYS_USE_CPU or len(detectors) == 0:
        detectors += _ref.create_object_detectors(delegate_class, stop_event, log_queue, frame_queue,
                                                  frame_buffers, model_path, kwargs)
    return detectors
