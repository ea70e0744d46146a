"""Detector factory + worker for the MI355X family: a superset of `watsor/detection/detector.py`.

`create_object_detectors` keeps the reference's signature and file-gate convention
(`detector.py:12-55`: `edgetpu.tflite` -> Coral, `gpu.trt` -> CUDA, else CPU) and adds one more
gate in front: `mi355x.bin` in the model path -> one detector process per AMD GPU.  When that file
is absent (or no AMD GPU / HIP library is present) it defers to the reference factory unchanged.

`BatchedObjectDetector` is the reference's `ObjectDetector` worker (`detector.py:58-112`) with these
differences, all inside the worker process:

  * after the blocking `get(timeout=1)` it drains up to `max_batch - 1` more payloads with `get_nowait()` and
    runs them as ONE batch.  (`BalancedQueue` holds at most one queued frame per camera, `watsor/stream/sync.py:156-166`,
    so a batch is one frame of each of several cameras: batch = 8 needs >= 8 cameras on the GPU.)
  * with a detector that offers `submit_host` / `collect` (HipObjectDetector) a batch is SUBMITTED, not waited for: the
    frames' pixels go from the shared-memory `Frame.image` arrays -- page-locked once with `wz_host_register`
    (`watsor/stream/share.py:35-41` allocates them from `multiprocessing` heap arenas) -- to the GPU by DMA on the
    lane's stream while the previous batches' kernels run; the worker keeps up to `lanes` batches in flight and
    retires the oldest (`wz_collect` writes the 100 `Detection` rows into `frame.header.detections`) when the lanes
    are full or the queue is empty;
  * every frame is tagged with its camera's id (`payload.sender` -> index), and when the camera configurations are
    handed over (`kwargs['hip_cameras']`) their Confidence / Area / Mask filters are registered on the engine in
    this process (`HipCameraFilter`), so the rows come back with `zones[]` filled in -- in drop mode
    (`kwargs['hip_drop']`) failing rows come back as all-zero rows, which is what `hip_detection_sieve()` with a
    filter-less `HipTrackFilter()` expects;
  * exactly one `frame.latch.next()` per dequeued payload (`detector.py:111-112`), in `finally`, like the reference; a
    batch that fails is retried frame by frame so that one bad frame does not take its neighbours' detections with it;
  * `inference_time` receives the per-frame share of a batch's time (ms / n): the reference's `/metrics` derives
    `fps_max = 1000 / inference_time` from it (`watsor/main.py:242-251`).

This module needs the reference package (`watsor.stream`) at run time -- it is a drop-in for an
installed Watsor, see INTEGRATION.md.  Everything below `detect_batch()` does not.
"""
from __future__ import annotations

from collections import deque
from os import path
from queue import Empty
from time import perf_counter

from numpy import uint8

from watsor_amd.detection.devices import hip_gpus
from watsor_amd.detection.hip_gpu import ENGINE_FILE

_ref = None        # watsor.detection.detector, bound on first use (the reference runtime is absent on a bare GPU box)
_worker_class = None


def _require_watsor():
    """Binds the reference runtime the first time something needs it (not at import: `BatchedWorkerMixin` and
    `hip_detector_options` work without an installed Watsor)."""
    global _ref
    if _ref is None:
        try:
            from watsor.detection import detector as ref
        except ImportError as e:
            raise ImportError("watsor_amd.detection.detector plugs into an installed Watsor "
                              "(watsor.detection.detector / watsor.stream); it was not found on sys.path") from e
        _ref = ref
    return _ref


class BatchedWorkerMixin:
    """The batching / camera-binding / asynchronous logic of the worker, independent of the reference's base class
    (tests drive it with stand-ins for `Frame` / `FrameBuffer` where no Watsor is installed, e.g. on the GPU box)."""

    # -- per-process state, created on the first _process() call inside the worker ------------------------------
    def _hip_state(self, frame_buffers, object_detector, kwargs):
        st = getattr(self, "_hip_worker_state", None)
        if st is not None and st["detector"] is object_detector:
            return st
        st = dict(detector=object_detector, inflight=deque(), next_lane=0, cams=None, lanes=1, asynchronous=False)
        bind = getattr(object_detector, "bind_cameras", None)
        if bind is not None:
            st["cams"] = bind(frame_buffers, kwargs.get("hip_cameras"), bool(kwargs.get("hip_drop", False)),
                              logger=getattr(self, "_logger", None))
        if hasattr(object_detector, "submit_host") and hasattr(object_detector, "collect"):
            st["asynchronous"] = kwargs.get("hip_async", True)
            st["lanes"] = max(1, min(int(kwargs.get("hip_lanes", 2)), getattr(object_detector, "num_lanes", 1)))
        self._hip_worker_state = st
        return st

    def _process(self, frame_queue, stop_event, frame_buffers, fps, inference_time, object_detector, *args, **kwargs):
        st = self._hip_state(frame_buffers, object_detector, kwargs)
        payloads = []
        try:
            # block for a frame only when nothing is in flight: a batch on the GPU is retired as soon as the queue runs dry
            first = frame_queue.get(timeout=1) if not st["inflight"] else frame_queue.get_nowait()
            if first is not None:
                payloads.append(first)
        except Empty:
            pass
        if payloads and hasattr(object_detector, "detect_batch"):
            limit = getattr(object_detector, "max_batch", 1)
            while len(payloads) < limit:
                try:
                    nxt = frame_queue.get_nowait()
                except Empty:
                    break
                if nxt is not None:
                    payloads.append(nxt)
        if payloads:
            if st["asynchronous"]:
                self._submit_frames(st, payloads, frame_buffers, fps, inference_time, object_detector)
            else:
                self._next_frames(payloads, stop_event, frame_buffers, fps, inference_time, object_detector, st)
        if st["inflight"] and (not payloads or len(st["inflight"]) >= st["lanes"]):
            self._retire_oldest(st, fps, inference_time, object_detector)
        elif not payloads:
            return self._no_frame(stop_event, frame_buffers, fps, inference_time, object_detector, *args, **kwargs)

    # -- synchronous path (any plugin with detect(); detect_batch() when it has one) --------------------------------
    def _resolve(self, payloads, frame_buffers):
        """[(payload, frame)] of the payloads that name a frame of a known camera; the others are dropped with a warning
        (there is no frame whose latch could be stepped for them)."""
        out = []
        for p in payloads:
            try:
                out.append((p, frame_buffers[p.sender].frames[p.frame_index]))
            except (KeyError, IndexError, TypeError):
                self._warn("payload of unknown sender / frame %r dropped" % (p,))
        return out

    def _next_frames(self, payloads, stop_event, frame_buffers, fps, inference_time, object_detector, st=None):
        frames = self._resolve(payloads, frame_buffers)
        try:
            done = False
            if len(frames) > 1 and hasattr(object_detector, "detect_batch"):
                try:
                    shapes, images, rows = [], [], []
                    for _, frame in frames:
                        image_shape, image_np = frame.get_numpy_image(uint8)
                        shapes.append(image_shape)
                        images.append(image_np)
                        rows.append(frame.header.detections)
                    cams = self._camera_ids(st, [p for p, _ in frames])
                    if cams is not None:
                        time_of_inference = object_detector.detect_batch(shapes, images, rows, cameras=cams)
                    else:
                        time_of_inference = object_detector.detect_batch(shapes, images, rows)
                    for _ in frames:
                        inference_time(value=time_of_inference / len(frames))
                        fps(value=True)
                    done = True
                except ValueError as e:           # a frame the engine cannot take (size): the others still get detected
                    self._warn("batch of %d frames rejected (%s), retrying frame by frame" % (len(frames), e))
            if not done:
                for p, frame in frames:
                    try:
                        self._detect_one(st, p, frame, fps, inference_time, object_detector)
                    except ValueError as e:
                        self._warn("frame of %s skipped: %s" % (p.sender, e))
        finally:
            for _, frame in frames:               # one latch step per dequeued payload, whatever happened above
                frame.latch.next()

    def _detect_one(self, st, payload, frame, fps, inference_time, object_detector):
        image_shape, image_np = frame.get_numpy_image(uint8)
        cams = self._camera_ids(st, [payload])
        if cams is not None and hasattr(object_detector, "detect_batch"):
            t = object_detector.detect_batch([image_shape], [image_np], [frame.header.detections], cameras=cams)
        else:
            t = object_detector.detect(image_shape, image_np, frame.header.detections)
        inference_time(value=t)
        fps(value=True)

    @staticmethod
    def _camera_ids(st, payloads):
        if st is None or st["cams"] is None:
            return None
        return [st["cams"].get(p.sender, -1) for p in payloads]

    def _warn(self, msg):
        logger = getattr(self, "_logger", None)
        if logger is not None:
            logger.warning(msg)

    # -- asynchronous path: submit now, retire later ------------------------------------------------------------------
    def _submit_frames(self, st, payloads, frame_buffers, fps, inference_time, object_detector):
        resolved = self._resolve(payloads, frame_buffers)
        if not resolved:
            return
        payloads = [p for p, _ in resolved]
        frames = [f for _, f in resolved]
        try:
            images = [f.get_numpy_image(uint8)[1] for f in frames]
            lane = st["next_lane"]
            t0 = perf_counter()
            object_detector.submit_host(lane, images, self._camera_ids(st, payloads))
        except Exception as e:                    # nothing was enqueued for this batch: synchronous retry, frame by frame
            self._warn("asynchronous submit of %d frames failed (%s), falling back to synchronous calls" % (len(payloads), e))
            # (the lane this batch would have used may still hold an older batch: retire everything first, in order)
            while st["inflight"]:
                self._retire_oldest(st, fps, inference_time, object_detector)
            return self._next_frames(payloads, None, frame_buffers, fps, inference_time, object_detector, st)
        st["inflight"].append((lane, frames, t0))
        st["next_lane"] = (lane + 1) % st["lanes"]

    def _retire_oldest(self, st, fps, inference_time, object_detector):
        lane, frames, t0 = st["inflight"].popleft()
        try:
            object_detector.collect(lane, [f.header.detections for f in frames])
            ms = (perf_counter() - t0) * 1000.0
            for _ in frames:
                inference_time(value=ms / len(frames))
                fps(value=True)
        finally:
            for f in frames:
                f.latch.next()

    def drain(self, fps=None, inference_time=None):
        """Retire everything still in flight (worker shutdown)."""
        st = getattr(self, "_hip_worker_state", None)
        noop = lambda **kw: None                                                  # noqa: E731
        while st is not None and st["inflight"]:
            self._retire_oldest(st, fps or noop, inference_time or noop, st["detector"])


def _batched_object_detector():
    """`BatchedObjectDetector`: the mixin on top of the reference's own `ObjectDetector` (derived on first use)."""
    global _worker_class
    if _worker_class is None:
        ref = _require_watsor()

        class BatchedObjectDetector(BatchedWorkerMixin, ref.ObjectDetector):
            """`ObjectDetector` that dequeues several frames and runs one batched (asynchronous) detection."""

            def _spin(self, action, stop_event, *args, **kwargs):      # (a staticmethod in the reference: spin.py:51-54)
                try:
                    while not stop_event.is_set():
                        action(*args, **kwargs)
                finally:
                    try:
                        self.drain()              # rows of the batches still on the GPU, before the engine goes away
                    finally:
                        self._hip_worker_state = None

        BatchedObjectDetector.__module__ = __name__
        BatchedObjectDetector.__qualname__ = "BatchedObjectDetector"
        _worker_class = BatchedObjectDetector
    return _worker_class


def __getattr__(name):        # `from watsor_amd.detection.detector import BatchedObjectDetector`
    if name == "BatchedObjectDetector":
        return _batched_object_detector()
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def hip_detector_options(frame_buffers, kwargs):
    """The third positional argument of `HipObjectDetector`: what the engine has to reserve, derived from the cameras
    this process is given (the largest frame of any `FrameBuffer`, `watsor/stream/share.py:27-31`) instead of from
    environment defaults -- a 4K camera must not kill the worker with WZ_ELIMIT on its first frame."""
    opts = dict(kwargs.get("hip_options") or {})
    widths, heights = [], []
    for fb in frame_buffers.values():
        for frame in fb.frames[:1]:
            widths.append(int(frame.header.width))
            heights.append(int(frame.header.height))
    if widths:
        opts.setdefault("max_width", max(widths))
        opts.setdefault("max_height", max(heights))
    return opts


def create_object_detectors(delegate_class, stop_event, log_queue, frame_queue, frame_buffers, model_path,
                            kwargs=None):
    """Creates all available detectors: AMD GPUs first when `mi355x.bin` is provided, otherwise
    whatever the reference factory finds (Coral, CUDA, CPU).  Same arguments and return value as
    `watsor.detection.detector.create_object_detectors` (detector.py:12-55).

    Optional entries of `kwargs` (all consumed inside the detector processes):
      hip_cameras  {camera name: normalised camera config}  -> the camera's Confidence / Area / Mask filters run on the GPU
      hip_drop     True: rows failing those filters come back as all-zero rows (for `hip_detection_sieve()`)
      hip_options  dict(max_batch=, max_width=, max_height=) overriding what is derived from the frame buffers;
                   pixel_format= "rgb24" | "nv12" | "yuv420p" (or {camera name: ...}): what the decoders write (hip_gpu.py)
      hip_lanes    batches kept in flight per GPU by the worker (default 2)"""
    _ref = _require_watsor()
    detectors = []
    if kwargs is None:
        kwargs = {}

    if path.isfile(path.join(model_path, ENGINE_FILE)):
        options = hip_detector_options(frame_buffers, kwargs)
        BatchedObjectDetector = _batched_object_detector()
        for device, clazz in hip_gpus():
            detectors.append(BatchedObjectDetector(
                delegate_class, "detector{}".format(len(detectors) + 1), stop_event, log_queue, frame_queue,
                frame_buffers, kwargs={**kwargs, 'detector_class': clazz, 'detector_args': (model_path, device, options)}))

    if _ref._ALWAYS_USE_CPU or len(detectors) == 0:
        detectors += _ref.create_object_detectors(delegate_class, stop_event, log_queue, frame_queue,
                                                  frame_buffers, model_path, kwargs)
    return detectors
