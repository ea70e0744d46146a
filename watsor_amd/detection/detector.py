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
  * `inference_time` receives the per-frame share of a batch's SERVICE time (submit, or the previous retirement if later,
    to collect; ms / n), one observation per batch or per 5 ms of batches: the reference's `/metrics` derives
    `fps_max = 1000 / inference_time` from it (`watsor/main.py:242-251`) and the gauge is a mean over its observations
    (`watsor/stream/share.py:225-238`);
  * with `HipObjectDetector` every Frame of every FrameBuffer is described to the engine once (`wz_bind_frames`), so a
    batch costs the worker one call with a list of table indices and one call to collect.

This module needs the reference package (`watsor.stream`) at run time -- it is a drop-in for an
installed Watsor, see INTEGRATION.md.  Everything below `detect_batch()` does not.
"""
from __future__ import annotations

import os
from collections import deque
from os import path
from queue import Empty
from time import perf_counter

from numpy import uint8

from watsor_amd._lib import RowsIncomplete
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
    (tests drive it with stand-ins for `Frame` / `FrameBuffer` where no Watsor is installed, e.g. on the GPU box).

    Per dequeued payload the worker itself does one dictionary lookup and two list appends: with a detector that offers
    `bind_frame_table` (HipObjectDetector) every Frame is described to the engine once, a batch is a list of table indices
    handed over in ONE call (`submit_bound`), the rows land in the frames' own headers in ONE call (`collect_bound`).  What is
    left per frame is the reference runtime's own: `Queue.get`, `latch.next()`, `fps(value=True)`."""

    # -- per-process state, created on the first _process() call inside the worker ------------------------------
    def _hip_state(self, frame_buffers, object_detector, kwargs):
        st = getattr(self, "_hip_worker_state", None)
        if st is not None and st["detector"] is object_detector:
            return st
        st = dict(detector=object_detector, inflight=deque(), next_lane=0, cams=None, lanes=1, asynchronous=False, table=None, affinity=None,
                  limit=getattr(object_detector, "max_batch", 1) if hasattr(object_detector, "detect_batch") else 1,
                  last_retire=0.0, acc_ms=0.0, acc_frames=0, last_flush=0.0,
                  metric_interval=float(kwargs.get("hip_metric_interval", 0.005)))
        aff = kwargs.get("hip_affinity")
        if isinstance(aff, dict) and aff.get("count", 1) > 1:
            # camera affinity (`create_object_detectors(..., kwargs={'hip_affinity': True})`): this detector owns the cameras the factory dealt
            # to it and binds / page-locks THEIR frame buffers only; payloads of other cameras that it draws from the shared queue are
            # handed to their owner's side queue (`_gather_with_affinity`)
            st["affinity"] = aff
            frame_buffers = {n: fb for n, fb in frame_buffers.items() if aff["owners"].get(n, aff["index"]) == aff["index"]}
            st["own_cameras"] = sorted(str(n) for n in frame_buffers)
            self._info("camera affinity: detector %d of %d serves %d cameras (%s)" % (
                aff["index"] + 1, aff["count"], len(frame_buffers), ", ".join(st["own_cameras"][:8]) + (" ..." if len(frame_buffers) > 8 else "")))
        bind = getattr(object_detector, "bind_cameras", None)
        if bind is not None:
            st["cams"] = bind(frame_buffers, kwargs.get("hip_cameras"), bool(kwargs.get("hip_drop", False)),
                              logger=getattr(self, "_logger", None))
        if hasattr(object_detector, "submit_host") and hasattr(object_detector, "collect"):
            st["asynchronous"] = kwargs.get("hip_async", True)
            lanes = getattr(object_detector, "num_lanes", 1)
            st["lanes"] = max(1, min(int(kwargs.get("hip_lanes", lanes)), lanes))
        if st["asynchronous"] and kwargs.get("hip_frame_table", True) and hasattr(object_detector, "bind_frame_table"):
            try:
                st["table"] = object_detector.bind_frame_table(frame_buffers, st["cams"] or {})
            except ValueError as e:               # a frame buffer the engine cannot take: per-batch descriptions, which skip only its frames
                self._warn("frame table not bound (%s): frames are described per batch" % (e,))
        self._hip_worker_state = st
        return st

    def _process(self, frame_queue, stop_event, frame_buffers, fps, inference_time, object_detector, *args, **kwargs):
        st = getattr(self, "_hip_worker_state", None)
        if st is None or st["detector"] is not object_detector:
            st = self._hip_state(frame_buffers, object_detector, kwargs)
        inflight = st["inflight"]
        payloads = []
        if st["affinity"] is not None:
            payloads = self._gather_with_affinity(st, frame_queue, block=not inflight)
        else:
            try:
                # block for a frame only when nothing is in flight: a batch on the GPU is retired as soon as the queue runs dry
                first = frame_queue.get(timeout=1) if not inflight else frame_queue.get_nowait()
                if first is not None:
                    payloads.append(first)
                    get_nowait, limit = frame_queue.get_nowait, st["limit"]
                    while len(payloads) < limit:
                        nxt = get_nowait()
                        if nxt is not None:
                            payloads.append(nxt)
            except Empty:
                pass
        if payloads:
            if st["asynchronous"]:
                self._submit_frames(st, payloads, frame_buffers, fps, inference_time, object_detector)
            else:
                self._next_frames(payloads, stop_event, frame_buffers, fps, inference_time, object_detector, st)
        if inflight and (not payloads or len(inflight) >= st["lanes"]):
            self._retire_oldest(st, fps, inference_time, object_detector)
        elif not payloads:
            if st["acc_frames"]:
                self._observe(st, inference_time, 0.0, 0, perf_counter(), flush=True)
            return self._no_frame(stop_event, frame_buffers, fps, inference_time, object_detector, *args, **kwargs)

    # -- camera affinity: one shared queue, one owner per camera -------------------------------------------------------
    AFFINITY_POLL_S = 0.002

    def _gather_with_affinity(self, st, frame_queue, block):
        """The payloads of THIS detector's cameras, up to the batch limit: first what the other detectors handed over (side queue), then
        the shared queue -- whose `get` releases the camera's semaphore exactly as in the reference (`watsor/stream/sync.py:156-166`), so
        the decoder may queue its next frame -- keeping what is ours and passing the rest to its owner's side queue.  Nothing is
        latched here: the owner steps the latch of every payload it processes, once.  With nothing in flight the shared queue is
        polled every AFFINITY_POLL_S (two sources cannot both be blocked on)."""
        aff = st["affinity"]
        me, owners, sides = aff["index"], aff["owners"], aff["side"]
        mine, limit = [], st["limit"]
        side = sides[me]
        try:
            while len(mine) < limit:
                mine.append(side.get_nowait())
        except Empty:
            pass
        first = True
        while len(mine) < limit:
            try:
                p = frame_queue.get(timeout=self.AFFINITY_POLL_S) if (first and block and not mine) else frame_queue.get_nowait()
            except Empty:
                break
            first = False
            if p is None:
                continue
            k = owners.get(getattr(p, "sender", None), me)      # (an unknown sender stays here and is dropped with the usual warning)
            if k == me:
                mine.append(p)
            else:
                sides[k].put(p)
                st["forwarded"] = st.get("forwarded", 0) + 1
        return mine

    def _info(self, msg):
        logger = getattr(self, "_logger", None)
        if logger is not None:
            logger.info(msg)

    # -- synchronous path (any plugin with detect(); detect_batch() when it has one) --------------------------------
    def _resolve(self, payloads, frame_buffers):
        """[(payload, frame)] of the payloads that name a frame of a known camera; the others are dropped with a warning
        (there is no frame whose latch could be stepped for them)."""
        out = []
        for p in payloads:
            try:
                if p.frame_index < 0:
                    raise IndexError(p.frame_index)
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
                    t0 = perf_counter()
                    if cams is not None:
                        time_of_inference = object_detector.detect_batch(shapes, images, rows, cameras=cams)
                    else:
                        time_of_inference = object_detector.detect_batch(shapes, images, rows)
                    for _ in frames:
                        inference_time(value=time_of_inference / len(frames))
                        fps(value=True)
                    done = True
                except RowsIncomplete as e:       # rows written (and complete for every frame but the one named): no retry
                    self._warn("batch of %d frames: %s" % (len(frames), e))
                    share = (perf_counter() - t0) * 1000.0 / len(frames)   # (the call raised instead of returning its time: the
                    for _ in frames:                                        # gauge must not go stale under a persistent overflow)
                        inference_time(value=share)
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
        t0 = perf_counter()
        try:
            if cams is not None and hasattr(object_detector, "detect_batch"):
                t = object_detector.detect_batch([image_shape], [image_np], [frame.header.detections], cameras=cams)
            else:
                t = object_detector.detect(image_shape, image_np, frame.header.detections)
        except RowsIncomplete as e:               # the rows were written (possibly fewer than the frame deserves): it counts
            self._warn("frame of %s: %s" % (payload.sender, e))
            inference_time(value=(perf_counter() - t0) * 1000.0)
            fps(value=True)
            return
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
        table = st["table"]
        frames = None
        if table is not None:
            # bound frames: payload -> table index; nothing of the frame itself is touched here
            kept, entries, latches = [], [], []
            for p in payloads:
                try:
                    base, nexts = table[p.sender]
                    i = p.frame_index
                    if i < 0:
                        raise IndexError(i)
                    latches.append(nexts[i])
                    entries.append(base + i)
                    kept.append(p)
                except (KeyError, IndexError, TypeError):
                    self._warn("payload of unknown sender / frame %r dropped" % (p,))
            payloads = kept
        else:
            resolved = self._resolve(payloads, frame_buffers)
            payloads = [p for p, _ in resolved]
            frames = [f for _, f in resolved]
            latches = [f.latch.next for f in frames]
        if not payloads:
            return
        try:
            lane = st["next_lane"]
            t0 = perf_counter()
            if frames is None:
                object_detector.submit_bound(lane, entries)
            else:
                object_detector.submit_host(lane, [f.get_numpy_image(uint8)[1] for f in frames], self._camera_ids(st, payloads))
        except Exception as e:                    # nothing was enqueued for this batch: synchronous retry, frame by frame
            self._warn("asynchronous submit of %d frames failed (%s), falling back to synchronous calls" % (len(payloads), e))
            # (the lane this batch would have used may still hold an older batch: retire everything first, in order)
            while st["inflight"]:
                self._retire_oldest(st, fps, inference_time, object_detector)
            return self._next_frames(payloads, None, frame_buffers, fps, inference_time, object_detector, st)
        st["inflight"].append((lane, latches, t0, frames))
        st["next_lane"] = (lane + 1) % st["lanes"]

    def _retire_oldest(self, st, fps, inference_time, object_detector):
        lane, latches, t0, frames = st["inflight"].popleft()
        try:
            try:
                if frames is None:
                    object_detector.collect_bound(lane)
                else:
                    object_detector.collect(lane, [f.header.detections for f in frames])
            except RowsIncomplete as e:           # the rows WERE written; the engine says what is wrong with them.  Any other
                self._warn("batch of %d frames: %s" % (len(latches), e))   # failure (rows not written) propagates: detector.py:99-100
            now = perf_counter()
            # the batch's SERVICE time: from its submit, or from the previous retirement when it was queued behind another
            # lane's batch until then -- what `fps_max = 1000 / inference_time` (watsor/main.py:242-251) should be derived from
            self._observe(st, inference_time, (now - max(t0, st["last_retire"])) * 1000.0, len(latches), now)
            st["last_retire"] = now
            for _ in latches:
                fps(value=True)
        finally:
            for step in latches:                  # one latch step per dequeued payload, whatever happened above
                step()

    @staticmethod
    def _observe(st, inference_time, ms, n, now, flush=False):
        """`inference_time` is a mean over its observations (`watsor/stream/share.py:225-238` -- a Python loop over 100 shared
        cells per call, ~120 us): n frames of one batch would be n identical observations, and back-to-back batches are folded
        into one observation of their mean per `hip_metric_interval` seconds (default 5 ms; 0: one per batch)."""
        st["acc_ms"] += ms
        st["acc_frames"] += n
        if st["acc_frames"] and (flush or now - st["last_flush"] >= st["metric_interval"]):
            inference_time(value=st["acc_ms"] / st["acc_frames"])
            st["acc_ms"], st["acc_frames"], st["last_flush"] = 0.0, 0, now

    def drain(self, fps=None, inference_time=None):
        """Retire everything still in flight (worker shutdown)."""
        st = getattr(self, "_hip_worker_state", None)
        noop = lambda **kw: None                                                  # noqa: E731
        while st is not None and st["inflight"]:
            self._retire_oldest(st, fps or noop, inference_time or noop, st["detector"])
        if st is not None and st["acc_frames"]:
            self._observe(st, inference_time or noop, 0.0, 0, perf_counter(), flush=True)


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


def hip_detector_options(frame_buffers, kwargs, n_detectors=1):
    """The third positional argument of `HipObjectDetector`: what the engine has to reserve, derived from the cameras
    this process is given (the largest frame of any `FrameBuffer`, `watsor/stream/share.py:27-31`) instead of from
    environment defaults -- a 4K camera must not kill the worker with WZ_ELIMIT on its first frame.
    `n_detectors`: AMD detector processes that share these cameras' one queue (`watsor/main.py:414-418`): batch size and schedule
    follow the cameras PER DETECTOR, ceil(cameras / detectors) -- 16 cameras on an 8-GPU host are two per detector, not sixteen."""
    opts = dict(kwargs.get("hip_options") or {})
    per_detector = -(-len(frame_buffers) // max(1, int(n_detectors)))
    widths, heights = [], []
    for fb in frame_buffers.values():
        for frame in fb.frames[:1]:
            widths.append(int(frame.header.width))
            # a one-"channel" buffer holds the (H * 3 / 2, W) bytes of an NV12 / yuv420p picture: the engine is sized for H
            planar = int(frame.header.channels) == 1 and int(frame.header.height) % 3 == 0
            heights.append(int(frame.header.height) // 3 * 2 if planar else int(frame.header.height))
    if widths:
        opts.setdefault("max_width", max(widths))
        opts.setdefault("max_height", max(heights))
    # More than 8 cameras: batches of up to 16.  The worker never WAITS for a batch to fill (it takes what is queued), so the limit
    # only matters once that many frames are waiting -- and then one batch of 16 is both faster and sooner done than two of 8
    # (16 cameras, one MI355X: 40.1 k frames/s at 0.24 ms enqueue-to-latch against 34.1 k at 1.31 ms).
    # An operator's own setting -- `hip_options["max_batch"]` or WATSOR_HIP_MAX_BATCH -- is left alone.
    if per_detector > 8 and "max_batch" not in opts and not os.environ.get("WATSOR_HIP_MAX_BATCH"):
        opts["max_batch"] = 16
    # The schedule: "auto" (or nothing) -> by the number of cameras this detector serves.  `BalancedQueue` holds one queued frame per
    # camera (`watsor/stream/sync.py:156-166`), so with 2 - 4 cameras a batch is 2 - 4 frames and at most one or two are in flight:
    # the launch shapes that finish a lone batch soonest are the right ones (a lone batch of four host frames 0.434 -> 0.394 ms); with
    # more cameras the lanes fill and the throughput shapes give 5 - 9 % more frames per second.  ONE camera:
    # a lone frame is launched kernel by kernel under either schedule and takes the same time (p50 0.286 ms both ways), while the
    # throughput shapes carry 27 % more frames when the camera is faster than the detector (11 765 against 9 284 frames/s,
    # bench.py legs.latency_schedule_b8 / config3_1x720p_b1) -- so one camera gets throughput.  The decision is a PREFERENCE
    # ("auto:<name>"): a process whose schedule is already fixed keeps it (HipObjectDetector logs the fact instead of raising).
    if str(opts.get("schedule") or "auto").lower() == "auto" and not os.environ.get("WZ_SCHEDULE"):
        opts["schedule"] = "auto:latency" if LATENCY_SCHEDULE_MIN_CAMERAS <= per_detector <= LATENCY_SCHEDULE_MAX_CAMERAS else "auto:throughput"
    return opts


LATENCY_SCHEDULE_MIN_CAMERAS = 2
LATENCY_SCHEDULE_MAX_CAMERAS = 4


def camera_affinity(frame_buffers, n_detectors, queue_factory=None):
    """Deals whole cameras to `n_detectors` detector processes, round-robin in the order of `frame_buffers` (the order of the
    configuration's camera list, `watsor/main.py:357-414`): {"count", "owners": {camera: detector index}, "side": one queue per
    detector}.  The reference has ONE shared queue for all detectors (`main.py:414-418`) and its decoders hold references to it, so
    the routing happens on the consumer side: a detector that draws another one's camera from the shared queue passes the payload to
    the owner's side queue (`BatchedWorkerMixin._gather_with_affinity`) -- one more hop for (n - 1) / n of the frames, in exchange
    for every frame buffer being page-locked and bound by ONE process, the one pinned to its GPU's NUMA node."""
    if queue_factory is None:
        import multiprocessing
        queue_factory = multiprocessing.Queue
    names = list(frame_buffers)
    return dict(count=int(n_detectors), owners={n: i % int(n_detectors) for i, n in enumerate(names)},
                side=[queue_factory() for _ in range(int(n_detectors))])


def create_object_detectors(delegate_class, stop_event, log_queue, frame_queue, frame_buffers, model_path,
                            kwargs=None):
    """Creates all available detectors: AMD GPUs first when `mi355x.bin` is provided, otherwise
    whatever the reference factory finds (Coral, CUDA, CPU).  Same arguments and return value as
    `watsor.detection.detector.create_object_detectors` (detector.py:12-55).

    Optional entries of `kwargs` (all consumed inside the detector processes):
      hip_cameras  {camera name: normalised camera config}  -> the camera's Confidence / Area / Mask filters run on the GPU
      hip_drop     True: rows failing those filters come back as all-zero rows (for `hip_detection_sieve()`)
      hip_options  dict(max_batch=, max_width=, max_height=) overriding what is derived from the frame buffers (the largest
                   frame; max_batch 16 for more than 8 cameras unless WATSOR_HIP_MAX_BATCH says otherwise, else the plugin's 8);
                   pixel_format= "rgb24" | "nv12" | "yuv420p" (or {camera name: ...}): what the decoders write (hip_gpu.py);
                   schedule= "latency" | "throughput" | "auto" (default: latency for up to 4 cameras per detector, WZ_SCHEDULE wins);
                   numa= True | False | "auto" (default: pin each detector process to its GPU's NUMA node on multi-GPU hosts)
      hip_lanes    batches kept in flight per GPU by the worker (default: the engine's lanes, 4)
      hip_metric_interval  seconds of batches folded into one `inference_time` observation (default 0.005; 0: one per batch)
      hip_frame_table      False: describe the frames of every batch to the engine instead of binding them once
      hip_affinity         True (several AMD GPUs): whole cameras are dealt to the detectors (`camera_affinity`): a detector binds and
                           page-locks only its own cameras' frame buffers; payloads drawn by the wrong detector are passed on"""
    _ref = _require_watsor()
    detectors = []
    if kwargs is None:
        kwargs = {}

    if path.isfile(path.join(model_path, ENGINE_FILE)):
        gpus = list(hip_gpus())
        BatchedObjectDetector = _batched_object_detector()
        affinity = camera_affinity(frame_buffers, len(gpus)) if (kwargs.get("hip_affinity") and len(gpus) > 1) else None
        for k, (device, clazz) in enumerate(gpus):
            kw = {key: v for key, v in kwargs.items() if key != "hip_affinity"}
            if affinity is not None:
                # whole cameras are dealt to the GPUs (north star: "whole cameras are hashed across the 8 GPUs"): detector k sizes its
                # engine for, binds and page-locks ITS cameras only
                own = {n: fb for n, fb in frame_buffers.items() if affinity["owners"][n] == k}
                options = hip_detector_options(own, kwargs, 1)
                kw["hip_affinity"] = dict(affinity, index=k)
            else:
                options = hip_detector_options(frame_buffers, kwargs, len(gpus))
            detectors.append(BatchedObjectDetector(
                delegate_class, "detector{}".format(len(detectors) + 1), stop_event, log_queue, frame_queue,
                frame_buffers, kwargs={**kw, 'detector_class': clazz, 'detector_args': (model_path, device, options)}))

    # The reference's own gates run as they always do (`detector.py:40-50`): Coral for edgetpu.tflite, CUDA for gpu.trt, CPU when
    # `_ALWAYS_USE_CPU` is set or NOTHING was found so far -- and the detectors they add continue the count ("detector3" after two AMD
    # GPUs: worker names stay unique, every family the model path provides for coexists on the one queue).
    others = _reference_detectors(_ref, delegate_class, stop_event, log_queue, frame_queue, frame_buffers, model_path, kwargs,
                                  first_index=len(detectors) + 1, found_so_far=len(detectors))
    return detectors + others


def _reference_detectors(ref, delegate_class, stop_event, log_queue, frame_queue, frame_buffers, model_path, kwargs, first_index,
                         found_so_far):
    """What `watsor.detection.detector.create_object_detectors` would append after `found_so_far` detectors already exist: its file
    gates and its CPU rule (`_ALWAYS_USE_CPU or len(detectors) == 0`, detector.py:48), named from `first_index` on.  The reference
    function itself cannot be told either (its `gen_name()` restarts at detector1 and its CPU rule counts only its own list)."""
    from watsor.detection.devices import cpus, cuda_gpus, edge_tpus
    out = []

    def append_detector(clazz, *args):
        out.append(ref.ObjectDetector(delegate_class, "detector{}".format(first_index + len(out)), stop_event, log_queue, frame_queue,
                                      frame_buffers, kwargs={**kwargs, 'detector_class': clazz, 'detector_args': (model_path, *args,)}))

    if path.isfile(path.join(model_path, 'edgetpu.tflite')):
        for device, clazz in edge_tpus():
            append_detector(clazz, device)
    if path.isfile(path.join(model_path, 'gpu.trt')):
        for device, clazz in cuda_gpus():
            append_detector(clazz, device)
    if ref._ALWAYS_USE_CPU or found_so_far + len(out) == 0:
        for clazz in cpus():
            append_detector(clazz)
    assert found_so_far + len(out) > 0, "Failed to create an object detector." \
                                        "Make sure TensorFlow is installed and model files are provided."
    return out
