"""The HIP plugin under the worker logic in a SPAWNED process (`watsor/main.py:474` sets the spawn start method;
`watsor/detection/detector.py:84-112` is the worker): HIP context created after spawn, frames and `Detection[100]` rows
in `multiprocessing.sharedctypes` memory created by the parent, that memory page-locked in the child
(`wz_host_register`), batches submitted asynchronously on two lanes -- and the rows the parent then reads are
bit-for-bit the rows of an in-process call on the same frames.  No Watsor needed (stand-ins: tests/shm_standins.py)."""
import ctypes

import numpy as np
import pytest

import shm_standins as shm
from conftest import make_engine
from watsor_amd.coco import COCO_CLASSES
from watsor_amd.runtime import ROW_DTYPE
from watsor_amd.synth import synthetic_frame, synthetic_zone_mask

pytestmark = pytest.mark.gpu


def rows_of(frame):
    return np.frombuffer(frame.header.get_obj().detections, dtype=ROW_DTYPE).copy()


def fill(frame, img):
    np.copyto(np.frombuffer(frame.image.get_obj(), np.uint8), img.reshape(-1))


def run_child(ctx, *args, **kw):
    import gpu_worker_child
    rq = ctx.Queue()
    args = tuple(str(a) if isinstance(a, str) else a for a in args)      # (conftest.ModelDir does not unpickle in the spawned child)
    p = ctx.Process(target=gpu_worker_child.run_worker, args=args + (rq,), kwargs=kw)
    p.start()
    status, a, b, c = rq.get(timeout=240)
    p.join(60)
    assert status == "ok", a
    assert p.exitcode == 0
    return a, b, c


@pytest.mark.parametrize("asynchronous", [True, False])
def test_plugin_in_a_spawned_worker_writes_the_same_rows(model_dir, asynchronous):
    ctx = shm.spawn_context()
    cams = {"cam%d" % c: shm.FrameBuffer(ctx, 3, *((640, 480) if c % 2 == 0 else (1280, 720))) for c in range(5)}
    images = {}
    batches = []
    for rnd in range(3):                                   # three rounds: one frame of every camera per round
        batch = []
        for c, (name, fb) in enumerate(sorted(cams.items())):
            img = synthetic_frame(fb.frames[0].header.width, fb.frames[0].header.height, 900 + 10 * rnd + c)
            fill(fb.frames[rnd], img)
            images[(name, rnd)] = img
            batch.append(shm.Payload(name, rnd))
        batches.append(batch)
    fps, inference_time = shm.Gauge(ctx), shm.Gauge(ctx)
    name, opts, pinned = run_child(ctx, model_dir, cams, batches, fps, inference_time, None, False, asynchronous=asynchronous)
    assert "gfx950" in name or "MI3" in name
    # derived from the frame buffers, not from env defaults; five cameras: the throughput schedule (up to four: latency -- the two-camera
    # tests below run their child on it and still compare bit for bit with this process's engine)
    assert opts == {"max_width": 1280, "max_height": 720, "schedule": "auto:throughput"}     # (a preference: watsor_amd/detection/detector.py)
    assert pinned == 15                                             # every Frame.image of every camera was page-locked
    # the parent reads the rows out of shared memory and compares with an in-process engine
    e = make_engine(model_dir, max_batch=8, max_width=1280, max_height=720)
    try:
        for rnd, batch in enumerate(batches):                      # the same batches (a batch size picks its split-K counts,
            refs = [np.zeros(100, ROW_DTYPE) for _ in batch]       # i.e. its fp32 summation order), in process
            e.detect_batch([images[(p.sender, rnd)] for p in batch], refs)
            for p, ref in zip(batch, refs):
                got = rows_of(cams[p.sender].frames[rnd])
                assert got.tobytes() == ref.tobytes(), (p.sender, rnd)
                assert got["label"][0] >= 1 and got["confidence"][0] > 0
    finally:
        e.close()
    for fb in cams.values():                                        # exactly one latch step per dequeued payload
        assert [f.latch.steps.value for f in fb.frames] == [1, 1, 1]
    # fps: one observation per frame.  inference_time: the per-frame share of a batch's service time -- per frame on the synchronous
    # path, one observation per batch (or per 5 ms of batches) on the asynchronous one (watsor_amd/detection/detector.py)
    assert fps.count.value == 15
    assert inference_time.count.value == 15 if not asynchronous else 1 <= inference_time.count.value <= 3
    assert 0 < inference_time.total.value / inference_time.count.value < 50


def test_spawned_worker_runs_the_camera_filters(model_dir):
    """`hip_cameras` / `hip_drop`: the camera's Confidence / Area / Mask filters are registered in the worker process and
    frames are tagged with their camera id -- rows failing the filters arrive as all-zero rows, the others carry zones."""
    ctx = shm.spawn_context()
    cams = {"porch": shm.FrameBuffer(ctx, 2, 640, 480), "yard": shm.FrameBuffer(ctx, 2, 640, 480)}
    alpha = synthetic_zone_mask(640, 480, 5, 4)
    import os
    import tempfile
    from PIL import Image
    d = tempfile.mkdtemp()
    rgba = np.zeros((480, 640, 4), np.uint8)
    rgba[..., 3] = alpha
    Image.fromarray(rgba, "RGBA").save(os.path.join(d, "porch.png"))
    cfg = {"width": 640, "height": 480, "mask": os.path.join(d, "porch.png"),
           "detect": [{n: {"area": 2, "confidence": 20, "zones": []}} for n in dict.fromkeys(COCO_CLASSES[1:])]}
    img = synthetic_frame(640, 480, 4242)
    for fb in cams.values():
        fill(fb.frames[0], img)
    batches = [[shm.Payload("porch", 0), shm.Payload("yard", 0)]]
    run_child(ctx, model_dir, cams, batches, shm.Gauge(ctx), shm.Gauge(ctx), {"porch": cfg}, True)
    porch, yard = rows_of(cams["porch"].frames[0]), rows_of(cams["yard"].frames[0])
    assert (yard["label"] >= 1).all() and not yard["zones"].any()               # no filter configured: raw rows
    dropped = porch["label"] == 0
    assert dropped.any() and not dropped.all()
    assert porch[dropped].tobytes() == bytes(72 * int(dropped.sum()))           # failing rows: all-zero records
    kept = ~dropped
    assert porch[kept].tobytes() != yard[kept].tobytes() or porch["zones"][kept].any()
    assert (porch["confidence"][kept] >= 0.2).all() and (porch["zones"][kept] > 0).any()
    # same verdict as the in-process filter on the raw rows
    from watsor_amd.filter.hip_filter import HipCameraFilter
    e = make_engine(model_dir, max_batch=2, max_width=640, max_height=480)
    try:
        flt = HipCameraFilter(e, 0, cfg)
        raw = yard.copy()
        ok = flt.filter_rows(raw).astype(bool)
        np.testing.assert_array_equal(ok, kept)
        np.testing.assert_array_equal(raw["zones"][kept], porch["zones"][kept])
    finally:
        e.close()


@pytest.mark.parametrize("asynchronous", [True, False])
def test_spawned_worker_takes_nv12_frame_buffers(model_dir, asynchronous):
    """SURVEY 8(f)-3 under the worker: a camera whose decoder writes NV12 has one-"channel" frame buffers of height * 3 / 2 rows
    (`share.py:37-40`), `hip_options["pixel_format"]` names it, and its rows equal an in-process call on the same bytes."""
    from oracle import yuv
    from watsor_amd.runtime import FMT_NV12, FMT_RGB24
    ctx = shm.spawn_context()
    cams = {"porch": shm.FrameBuffer(ctx, 2, 640, 480), "yard": shm.FrameBuffer(ctx, 2, 1280, 720 * 3 // 2, channels=1)}
    images, batches = {}, []
    for rnd in range(2):
        rgb = synthetic_frame(640, 480, 70 + rnd)
        nv = yuv.yuv420_from_rgb(synthetic_frame(1280, 720, 80 + rnd), "nv12")
        fill(cams["porch"].frames[rnd], rgb)
        fill(cams["yard"].frames[rnd], nv)
        images[("porch", rnd)], images[("yard", rnd)] = rgb, nv
        batches.append([shm.Payload("porch", rnd), shm.Payload("yard", rnd)])
    run_child(ctx, model_dir, cams, batches, shm.Gauge(ctx), shm.Gauge(ctx), None, False, asynchronous=asynchronous,
              hip_options={"pixel_format": {"yard": "nv12"}})
    e = make_engine(model_dir, max_batch=8, max_width=1280, max_height=1080)
    try:
        for rnd, batch in enumerate(batches):
            refs = [np.zeros(100, ROW_DTYPE) for _ in batch]
            e.detect_batch([images[(p.sender, rnd)] for p in batch], refs, formats=[FMT_RGB24, FMT_NV12])
            for p, ref in zip(batch, refs):
                got = rows_of(cams[p.sender].frames[rnd])
                assert got.tobytes() == ref.tobytes(), (p.sender, rnd)
            # the NV12 camera's boxes live in its picture's 1280 x 720 pixels, not in the buffer's 1080 rows
            yard = rows_of(cams["yard"].frames[rnd])
            assert yard["y_max"].max() <= 719 and yard["x_max"].max() <= 1279 and yard["confidence"][0] > 0
    finally:
        e.close()


def test_two_workers_share_one_queue_and_one_gpu(model_dir):
    """The reference's multi-device topology (`watsor/main.py:414-418`: every detector process pulls from the same `BalancedQueue`)
    with two `BatchedWorkerMixin` worker PROCESSES, here sharing the one GPU of the box: both page-lock and bind the same
    shared-memory frames, each drains up to max_batch payloads per turn.  Every dequeued payload is latched exactly once and
    counted once, both workers get work, no camera is starved, and the rows in shared memory are real detections.
    (CPU twin with the reference's own queue / sources / sieves: tests/test_two_workers_one_queue.py.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import worker_bench
    r = worker_bench.run(str(model_dir), n_cams=8, seconds=1.5, workers=2, gpus=1, costly=False, check=True, max_batch=4, warm_frames=20)
    print("\ntwo workers, one queue, one GPU: %s" % r)
    c = r["check"]
    assert c["latch_steps"] == c["fps_calls"] == c["worker_frames"] > 1000
    assert len(r["per_worker"]) == 2 and all(w["frames"] > 0.15 * c["worker_frames"] for w in r["per_worker"]), r["per_worker"]
    per_cam = c["per_camera_steps"]
    assert min(per_cam) > 0 and max(per_cam) <= 1.5 * min(per_cam) + 20, per_cam
    assert c["rows_written"] == c["frames_total"]                  # every frame of every camera carries detections
