"""`BatchedWorkerMixin` (watsor_amd/detection/detector.py) on the CPU with a scripted detector: batching, camera ids,
asynchronous submit / retire over the lanes, one latch step per payload, per-frame inference time, fall-backs."""
import queue

import numpy as np

import shm_standins as shm
from watsor_amd.detection.detector import BatchedWorkerMixin, hip_detector_options


class ScriptedDetector:
    max_batch = 4
    num_lanes = 3

    def __init__(self, reject_width=None):
        self.log = []
        self.busy = {}
        self.reject_width = reject_width
        self.max_in_flight = 0

    def bind_cameras(self, frame_buffers, camera_configs=None, drop=False, logger=None):
        self.log.append(("bind", sorted(frame_buffers), camera_configs, drop))
        return {n: i for i, n in enumerate(sorted(frame_buffers))}

    def _check(self, images):
        if self.reject_width is not None and any(im.shape[1] == self.reject_width for im in images):
            raise ValueError("frame too large")

    def _fill(self, im, det, cam):
        det[0].label = 1 + int(im.reshape(-1)[0])
        det[0].bounding_box.x_min = -1 if cam is None else cam

    def submit_host(self, lane, images, cameras=None):
        self._check(images)
        assert lane not in self.busy, "lane reused before it was collected"
        self.busy[lane] = (list(images), cameras)
        self.max_in_flight = max(self.max_in_flight, len(self.busy))
        self.log.append(("submit", lane, len(images)))

    def collect(self, lane, detections):
        images, cameras = self.busy.pop(lane)
        for i, (im, d) in enumerate(zip(images, detections)):
            self._fill(im, d, cameras[i] if cameras else None)
        self.log.append(("collect", lane, len(images)))

    def detect_batch(self, shapes, images, detections, cameras=None):
        self._check(images)
        for i, (im, d) in enumerate(zip(images, detections)):
            self._fill(im, d, cameras[i] if cameras else None)
        self.log.append(("sync", len(images)))
        return 4.0 * len(images)

    def detect(self, shape, image, detections):
        self._fill(image, detections, None)
        return 1.0


class Worker(BatchedWorkerMixin):
    _logger = None
    idle = 0

    def _no_frame(self, *a, **k):
        self.idle += 1


def setup(n_cams=3, wide=None):
    ctx = shm.spawn_context()
    cams = {"cam%d" % c: shm.FrameBuffer(ctx, 4, 96 if c == wide else 64, 48) for c in range(n_cams)}
    for c, fb in enumerate(sorted(cams)):
        for i, f in enumerate(cams[fb].frames):
            np.frombuffer(f.image.get_obj(), np.uint8)[:] = 10 * c + i
    return ctx, cams


def drive(w, det, cams, batches, **kwargs):
    ctx = shm.spawn_context()
    q = queue.Queue()
    fps, it = shm.Gauge(ctx), shm.Gauge(ctx)
    for b in batches:
        for p in b:
            q.put(p)
        while not q.empty():
            w._process(q, None, cams, fps, it, det, **kwargs)
    w.drain(fps, it)
    return fps, it


def first_row(frame):
    d = frame.header.get_obj().detections[0]
    return d.label, d.bounding_box.x_min


def test_async_batches_over_two_lanes():
    _, cams = setup()
    det, w = ScriptedDetector(), Worker()
    batches = [[shm.Payload("cam%d" % c, i) for c in range(3)] for i in range(4)]
    fps, it = drive(w, det, cams, batches, hip_lanes=2, hip_cameras={"cam1": {"x": 1}}, hip_drop=True, hip_metric_interval=0)
    assert det.log[0] == ("bind", ["cam0", "cam1", "cam2"], {"cam1": {"x": 1}}, True)
    assert [e for e in det.log if e[0] == "submit"] == [("submit", i % 2, 3) for i in range(4)]
    assert det.max_in_flight <= 2 and not det.busy
    for c in range(3):
        for i in range(4):
            f = cams["cam%d" % c].frames[i]
            assert first_row(f) == (1 + 10 * c + i, c)              # right frame, tagged with its camera id
            assert f.latch.steps.value == 1
    assert fps.count.value == 12 and it.count.value == 4          # fps per frame; inference_time once per batch (its per-frame share)


class TableDetector(ScriptedDetector):
    """... that also takes a frame table, like HipObjectDetector: a batch is then a list of table indices."""

    def bind_frame_table(self, frame_buffers, ids):
        self.entries, table = [], {}
        for name in sorted(frame_buffers):
            table[name] = (len(self.entries), [f.latch.next for f in frame_buffers[name].frames])
            self.entries += [(f, ids.get(name, -1)) for f in frame_buffers[name].frames]
        self.log.append(("table", len(self.entries)))
        return table

    def submit_bound(self, lane, entries):
        assert lane not in self.busy, "lane reused before it was collected"
        self.busy[lane] = list(entries)
        self.max_in_flight = max(self.max_in_flight, len(self.busy))
        self.log.append(("submit_bound", lane, list(entries)))

    def collect_bound(self, lane):
        for e in self.busy.pop(lane):
            f, cam = self.entries[e]
            self._fill(np.frombuffer(f.image.get_obj(), np.uint8), f.header.get_obj().detections, cam)
        self.log.append(("collect_bound", lane))


def test_bound_frames_travel_as_table_indices():
    _, cams = setup()
    det, w = TableDetector(), Worker()
    batches = [[shm.Payload("cam%d" % c, i) for c in range(3)] for i in range(4)]
    batches[1].append(shm.Payload("cam1", 9))          # no such frame
    batches[2].append(shm.Payload("stranger", 0))      # no such camera
    batches[3].append(shm.Payload("cam0", -1))         # a negative index is not "the last frame"
    fps, it = drive(w, det, cams, batches, hip_lanes=2)
    assert ("table", 12) in det.log and not [e for e in det.log if e[0] in ("submit", "sync")]
    subs = [e for e in det.log if e[0] == "submit_bound"]
    assert [e[1] for e in subs] == [0, 1, 0, 1] and subs[2][2] == [2, 6, 10]     # cam c, frame i -> 4 c + i
    assert det.max_in_flight <= 2 and not det.busy
    for c in range(3):
        for i in range(4):
            f = cams["cam%d" % c].frames[i]
            assert first_row(f) == (1 + 10 * c + i, c) and f.latch.steps.value == 1
    assert fps.count.value == 12 and 1 <= it.count.value <= 4       # observations folded per 5 ms by default


def test_frame_table_can_be_switched_off_and_a_refused_table_falls_back():
    _, cams = setup()
    det, w = TableDetector(), Worker()
    drive(w, det, cams, [[shm.Payload("cam0", 0), shm.Payload("cam1", 0)]], hip_frame_table=False)
    assert ("submit", 0, 2) in det.log and not [e for e in det.log if e[0] == "table"]

    class Refusing(TableDetector):
        def bind_frame_table(self, frame_buffers, ids):
            raise ValueError("camera 'cam1': frame memory smaller than its header says")

    det, w = Refusing(), Worker()
    drive(w, det, cams, [[shm.Payload("cam0", 1), shm.Payload("cam2", 1)]])
    assert ("submit", 0, 2) in det.log and first_row(cams["cam2"].frames[1]) == (22, 2)


def test_inference_time_is_service_time_not_queueing_latency():
    """Two batches in flight: the second one's observation starts at the first one's retirement, not at its own submit."""
    import time
    _, cams = setup()

    class Slow(TableDetector):
        max_batch = 2

        def collect_bound(self, lane):
            time.sleep(0.05)
            super().collect_bound(lane)

    det, w = Slow(), Worker()
    batches = [[shm.Payload("cam0", i), shm.Payload("cam1", i)] for i in range(3)]
    ctx = shm.spawn_context()
    q = queue.Queue()
    fps, it = shm.Gauge(ctx), shm.Gauge(ctx)
    for b in batches:
        for p in b:
            q.put(p)
    while not q.empty():
        w._process(q, None, cams, fps, it, det, hip_lanes=2, hip_metric_interval=0)
    w.drain(fps, it)
    # three batches of two frames, 50 ms each: every observation is ~25 ms per frame; latency-based it would grow (25, 50, ...)
    assert it.count.value == 3 and 60 < it.total.value < 100


def test_sync_path_reports_per_frame_time_and_camera_ids():
    _, cams = setup()
    det, w = ScriptedDetector(), Worker()
    fps, it = drive(w, det, cams, [[shm.Payload("cam0", 0), shm.Payload("cam2", 1)]], hip_async=False)
    assert ("sync", 2) in det.log and first_row(cams["cam2"].frames[1]) == (22, 2)
    assert it.count.value == 2 and abs(it.total.value - 8.0) < 1e-9   # 8 ms for the batch -> 4 ms per frame, twice
    assert w.idle == 0


def test_rejected_batch_is_retried_frame_by_frame_and_every_payload_is_released():
    _, cams = setup(wide=1)
    det, w = ScriptedDetector(reject_width=96), Worker()
    batch = [shm.Payload("cam0", 0), shm.Payload("cam1", 0), shm.Payload("cam2", 0), shm.Payload("nobody", 7)]
    fps, it = drive(w, det, cams, [batch], hip_lanes=2)
    assert first_row(cams["cam0"].frames[0]) == (1, 0) and first_row(cams["cam2"].frames[0]) == (21, 2)
    assert first_row(cams["cam1"].frames[0]) == (0, 0)                # the oversized camera's frame: skipped, not fatal
    assert [cams["cam%d" % c].frames[0].latch.steps.value for c in range(3)] == [1, 1, 1]
    assert fps.count.value == 2


def test_options_follow_the_frame_buffers(monkeypatch):
    monkeypatch.delenv("WATSOR_HIP_MAX_BATCH", raising=False)
    monkeypatch.delenv("WZ_SCHEDULE", raising=False)
    _, cams = setup(wide=2)
    assert hip_detector_options(cams, {}) == {"max_width": 96, "max_height": 48, "schedule": "auto:latency"}
    assert hip_detector_options(cams, {"hip_options": {"max_width": 4096, "max_batch": 16}}) == \
        {"max_width": 4096, "max_height": 48, "max_batch": 16, "schedule": "auto:latency"}
    assert hip_detector_options({}, {}) == {"schedule": "auto:throughput"}
    # more than 8 cameras: batches of up to 16 unless the installation says otherwise
    many = {"cam%d" % i: cams["cam0"] for i in range(9)}
    assert hip_detector_options(many, {})["max_batch"] == 16
    assert hip_detector_options(many, {"hip_options": {"max_batch": 4}})["max_batch"] == 4
    assert "max_batch" not in hip_detector_options({"cam%d" % i: cams["cam0"] for i in range(8)}, {})
    # ... and that includes the environment setting the plugin reads: the factory's 16 must not shadow it
    monkeypatch.setenv("WATSOR_HIP_MAX_BATCH", "12")
    assert "max_batch" not in hip_detector_options(many, {})
    assert hip_detector_options(many, {"hip_options": {"max_batch": 4}})["max_batch"] == 4


def test_schedule_is_chosen_by_the_number_of_cameras(monkeypatch):
    """`hip_options["schedule"]` = latency | throughput | auto.  auto (default): one queued frame per camera (`sync.py:156-166`) means
    two to four cameras never fill the lanes -- the launch shapes that finish a lone batch soonest; more cameras: throughput; ONE
    camera: throughput as well (a lone frame goes kernel by kernel under either schedule -- same p50 -- and the throughput shapes carry
    27 % more frames when the camera outruns the detector: bench.py legs).  What auto resolves to is a PREFERENCE
    ("auto:<name>": a process whose schedule is fixed keeps it); an explicit option or WZ_SCHEDULE in the environment wins."""
    monkeypatch.delenv("WZ_SCHEDULE", raising=False)
    _, cams = setup()
    n = lambda k: {"cam%d" % i: cams["cam0"] for i in range(k)}                  # noqa: E731
    assert hip_detector_options(n(1), {})["schedule"] == "auto:throughput"
    assert hip_detector_options(n(2), {})["schedule"] == "auto:latency"
    assert hip_detector_options(n(4), {})["schedule"] == "auto:latency"
    assert hip_detector_options(n(5), {})["schedule"] == "auto:throughput"
    assert hip_detector_options(n(4), {"hip_options": {"schedule": "auto"}})["schedule"] == "auto:latency"
    assert hip_detector_options(n(4), {"hip_options": {"schedule": "throughput"}})["schedule"] == "throughput"
    assert hip_detector_options(n(5), {"hip_options": {"schedule": "latency"}})["schedule"] == "latency"
    monkeypatch.setenv("WZ_SCHEDULE", "throughput")
    assert "schedule" not in hip_detector_options(n(4), {})                    # the operator's environment setting decides in the library


def test_options_count_cameras_per_detector(monkeypatch):
    """all detectors of a host pull from ONE queue (`watsor/main.py:414-418`), so what a detector sees is
    ceil(cameras / detectors): 16 cameras on 8 GPUs are two per detector (latency shapes, the plugin's batch of 8), 12 cameras on
    2 GPUs six each (throughput, 8), 20 on 2 ten each (throughput, 16)."""
    monkeypatch.delenv("WATSOR_HIP_MAX_BATCH", raising=False)
    monkeypatch.delenv("WZ_SCHEDULE", raising=False)
    _, cams = setup()
    n = lambda k: {"cam%d" % i: cams["cam0"] for i in range(k)}                  # noqa: E731
    o = hip_detector_options(n(16), {}, 8)
    assert o["schedule"] == "auto:latency" and "max_batch" not in o
    o = hip_detector_options(n(12), {}, 2)
    assert o["schedule"] == "auto:throughput" and "max_batch" not in o
    o = hip_detector_options(n(20), {}, 2)
    assert o["schedule"] == "auto:throughput" and o["max_batch"] == 16
    o = hip_detector_options(n(16), {})                                          # one detector: as before
    assert o["schedule"] == "auto:throughput" and o["max_batch"] == 16


class AffinityDetector(ScriptedDetector):
    """Remembers which cameras it was asked to bind and which frames it was given."""

    def __init__(self):
        super().__init__()
        self.bound, self.seen = None, []

    def bind_cameras(self, frame_buffers, camera_configs=None, drop=False, logger=None):
        self.bound = sorted(frame_buffers)
        return super().bind_cameras(frame_buffers, camera_configs, drop, logger)

    def submit_host(self, lane, images, cameras=None):
        self.seen += [int(im.reshape(-1)[0]) // 10 for im in images]
        super().submit_host(lane, images, cameras)


def test_camera_affinity_deals_whole_cameras_and_latches_exactly_once():
    """`kwargs['hip_affinity']` (north star: "whole cameras are hashed across the 8 GPUs"; the reference has ONE queue for all
    detectors, `watsor/main.py:414-418`): two workers on one shared queue, six cameras dealt round-robin.  Each worker binds only its
    own cameras' frame buffers, every payload is processed by its camera's owner -- whoever drew it from the shared queue -- and every
    payload's latch is stepped exactly once."""
    from watsor_amd.detection.detector import camera_affinity
    ctx, cams = setup(6)
    aff = camera_affinity(cams, 2, queue_factory=queue.Queue)
    assert aff["owners"] == {"cam0": 0, "cam1": 1, "cam2": 0, "cam3": 1, "cam4": 0, "cam5": 1} and len(aff["side"]) == 2
    shared = queue.Queue()
    workers = [Worker(), Worker()]
    dets = [AffinityDetector(), AffinityDetector()]
    gauges = [(shm.Gauge(ctx), shm.Gauge(ctx)) for _ in workers]
    payloads = [shm.Payload("cam%d" % c, i) for i in range(4) for c in range(6)]          # 24 payloads, every frame of every camera once
    for p in payloads:
        shared.put(p)
    for _ in range(200):                                                                   # the two workers take turns on the one queue
        for k, (w, d) in enumerate(zip(workers, dets)):
            w._process(shared, None, cams, gauges[k][0], gauges[k][1], d, hip_affinity=dict(aff, index=k))
        if shared.empty() and all(q.empty() for q in aff["side"]) and not any(w._hip_worker_state["inflight"] for w in workers):
            break
    for k, w in enumerate(workers):
        w.drain(*gauges[k])
    assert dets[0].bound == ["cam0", "cam2", "cam4"] and dets[1].bound == ["cam1", "cam3", "cam5"]      # disjoint camera sets
    assert sorted(set(dets[0].seen)) == [0, 2, 4] and sorted(set(dets[1].seen)) == [1, 3, 5]
    assert len(dets[0].seen) == 12 and len(dets[1].seen) == 12
    for c in range(6):
        for f in cams["cam%d" % c].frames:
            assert f.latch.steps.value == 1                                                # exactly once, by the owner
    assert sum(w._hip_worker_state.get("forwarded", 0) for w in workers) > 0              # (payloads really changed hands)
    assert gauges[0][0].count.value + gauges[1][0].count.value == 24


def test_camera_affinity_travels_into_spawned_workers():
    """The reference starts its detectors as spawned processes (`watsor/main.py:474`): what the factory puts into `kwargs['hip_affinity']` -- the
    owner table and one `multiprocessing.Queue` per detector -- is pickled into the child with the process's other arguments and must
    arrive alive: a payload put by the parent reaches detector 0's side queue in the child, and what the child forwards reaches detector 1's."""
    from watsor_amd.detection.detector import camera_affinity
    ctx = shm.spawn_context()
    _, cams = setup(4)
    aff = camera_affinity(cams, 2, queue_factory=ctx.Queue)
    out = ctx.Queue()
    child = ctx.Process(target=shm.affinity_echo, args=({"hip_affinity": dict(aff, index=0)}, out))
    child.start()
    try:
        aff["side"][0].put("payload-for-detector-0")
        assert out.get(timeout=60) == (0, 2, [("cam0", 0), ("cam1", 1), ("cam2", 0), ("cam3", 1)])
        assert aff["side"][1].get(timeout=20) == ("forwarded-by-0", "payload-for-detector-0")
    finally:
        child.join(30)
        if child.is_alive():
            child.terminate()
    assert child.exitcode == 0


def test_plain_plugin_without_batch_or_async_api():
    class Plain:
        def detect(self, shape, image, detections):
            detections[0].label = 5
            return 2.0

    _, cams = setup()
    w = Worker()
    fps, it = drive(w, Plain(), cams, [[shm.Payload("cam0", 0)], [shm.Payload("cam1", 1)]])
    assert first_row(cams["cam0"].frames[0])[0] == 5 and first_row(cams["cam1"].frames[1])[0] == 5
    assert fps.count.value == 2 and abs(it.total.value - 4.0) < 1e-9


def test_rows_incomplete_counts_the_batch_any_other_collect_failure_does_not(monkeypatch):
    """only WZ_EINCOMPLETE (`RowsIncomplete`) means "the rows were written" -- the batch is latched on and counted.
    Any other failure of collect (a bad slot, a lane without a bound batch: rows NOT written) must not pass as detected frames:
    it propagates like the reference's 'Detection failure' (detector.py:99-100); the latches are still stepped (detector.py:111-112).
    And on the synchronous path an incomplete batch is NOT re-run frame by frame."""
    import pytest
    from watsor_amd._lib import RowsIncomplete

    class Overflowing(ScriptedDetector):
        def collect(self, lane, detections):
            super().collect(lane, detections)
            raise RowsIncomplete("frame 1 of the batch on lane %d: rows may be incomplete" % lane)

    _, cams = setup()
    det, w = Overflowing(), Worker()
    fps, it = drive(w, det, cams, [[shm.Payload("cam%d" % c, 0) for c in range(3)]], hip_lanes=2, hip_metric_interval=0)
    assert fps.count.value == 3 and all(cams["cam%d" % c].frames[0].latch.steps.value == 1 for c in range(3))
    assert first_row(cams["cam1"].frames[0])[0] == 11                     # the rows are there

    class Broken(ScriptedDetector):
        def collect(self, lane, detections):
            self.busy.pop(lane)
            raise ValueError("wz_collect_bound: lane %d holds no bound batch" % lane)

    _, cams = setup()
    det, w = Broken(), Worker()
    with pytest.raises(ValueError, match="no bound batch"):
        drive(w, det, cams, [[shm.Payload("cam%d" % c, 0) for c in range(3)]], hip_lanes=2, hip_metric_interval=0)
    assert all(cams["cam%d" % c].frames[0].latch.steps.value == 1 for c in range(3))   # released all the same

    class SyncOverflowing(ScriptedDetector):
        def detect_batch(self, shapes, images, detections, cameras=None):
            super().detect_batch(shapes, images, detections, cameras)
            raise RowsIncomplete("frame 0: rows may be incomplete")

    _, cams = setup()
    det, w = SyncOverflowing(), Worker()
    ctx = shm.spawn_context()
    fps, it = shm.Gauge(ctx), shm.Gauge(ctx)
    w._next_frames([shm.Payload("cam%d" % c, 1) for c in range(3)], None, cams, fps, it, det)
    assert [e for e in det.log if e[0] == "sync"] == [("sync", 3)]         # one batched call, no frame-by-frame retry
    assert fps.count.value == 3 and all(cams["cam%d" % c].frames[1].latch.steps.value == 1 for c in range(3))
    assert it.count.value == 3 and it.total.value >= 0                     # ... and the gauge is fed although the call raised

    class OneOverflowing(ScriptedDetector):
        def detect(self, shape, image, detections):
            super().detect(shape, image, detections)
            raise RowsIncomplete("rows may be incomplete")

    _, cams = setup()
    det, w = OneOverflowing(), Worker()
    fps, it = shm.Gauge(ctx), shm.Gauge(ctx)
    w._next_frames([shm.Payload("cam0", 2)], None, cams, fps, it, det)     # one payload: the frame-by-frame path
    assert fps.count.value == 1 and it.count.value == 1 and cams["cam0"].frames[2].latch.steps.value == 1
