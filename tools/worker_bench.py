"""Frames/s through the detector WORKER LOOP in a spawned process (`bench.py` leg `worker_spawned`).

What a Watsor install sees of the engine is not `wz_submit_device` but `watsor/detection/detector.py:84-112` driven by
`watsor/stream/work.py:25-33`: `Queue.get` -> `frame_buffers[sender].frames[index]` -> detect -> `latch.next()`, one payload at
a time, in a process started with the spawn method (`watsor/main.py:474`).  Here the same loop runs `BatchedWorkerMixin` over
`HipObjectDetector` in a spawned process; cameras are shared-memory frame buffers created by the parent
(`multiprocessing.sharedctypes`, like `watsor/stream/share.py:35-41`), fed at saturation by producer processes through ONE
real `multiprocessing.Queue` under the reference's one-queued-frame-per-camera rule (`BalancedQueue`,
`watsor/stream/sync.py:144-166`).  No Watsor is installed on the GPU box: latch, gauges and the balanced queue are the
stand-ins of tests/shm_standins.py that take the same locks in the same order as the reference's classes (their per-call cost is
held against the reference's in tests/test_reference_plumbing.py).

Several workers on ONE queue -- the reference's multi-device topology (`watsor/main.py:414-418`: every detector process gets the
same `BalancedQueue`) -- with `workers=N`: N spawned worker processes, worker i on GPU `i % gpus`, i.e. `--workers 2 --gpus 1` is
two workers sharing one GPU, `--workers 8 --gpus 8` one worker per GPU of a node.  Reported per worker and summed.

    python tools/worker_bench.py [cameras] [seconds] [--workers N] [--gpus G] [--max-batch B]      # prints JSON objects
"""
import json
import os
import queue as pyqueue
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def producer(names, frame_buffers, q, sems, stop_event, ready):
    """Decoder stand-in: whenever one of its cameras has no frame queued it stamps the next frame of that camera's buffer
    (`header.epoch`, as `watsor/stream/ffmpeg.py:78-88` does) and enqueues its payload."""
    import shm_standins as shm
    bqs = {n: shm.BalancedQueueStandIn(q, sems, n) for n in names}
    nxt = {n: 0 for n in names}
    ready.set()
    while not stop_event.is_set():
        idle = True
        for n in names:
            fb = frame_buffers[n]
            try:
                i = nxt[n]
                fb.frames[i].header.epoch = time.time()
                bqs[n].put(shm.Payload(n, i), False)
                nxt[n] = (i + 1) % len(fb.frames)
                idle = False
            except pyqueue.Full:
                pass
        if idle:
            time.sleep(0.00005)


class NullDetector:
    """Harness self-test without a GPU (tests/test_worker_bench.py): takes the worker's calls, detects nothing.  Results
    obtained with it are marked invalid by `run`."""
    max_batch = 8
    num_lanes = 4
    device_name = "null (no GPU)"

    def __init__(self, model_dir, device, options):
        self.n = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def bind_cameras(self, frame_buffers, camera_configs=None, drop=False, logger=None):
        return {n: -1 for n in frame_buffers}

    def bind_frame_table(self, frame_buffers, ids):
        table, k = {}, 0
        for name in sorted(frame_buffers):
            table[name] = (k, [f.latch.next for f in frame_buffers[name].frames])
            k += len(frame_buffers[name].frames)
        return table

    def submit_bound(self, lane, entries):
        self.n[lane] = len(entries)

    def collect_bound(self, lane):
        time.sleep(float(os.environ.get('WZ_NULL_US_PER_FRAME', '200')) * 1e-6 * self.n.pop(lane))

    def submit_host(self, lane, images, cameras=None):
        self.n[lane] = len(images)

    def collect(self, lane, detections):
        time.sleep(float(os.environ.get('WZ_NULL_US_PER_FRAME', '200')) * 1e-6 * self.n.pop(lane))

    def detect_batch(self, *a, **k):
        return 1.0


def worker(model_dir, frame_buffers, q, sems, stop_event, fps, inference_time, result_q, kwargs, null_detector=False, device=0, tag=0):
    """`ObjectDetector._run` (`watsor/detection/detector.py:84-100`): plugin constructed in THIS process, then the spin loop."""
    try:
        import shm_standins as shm
        from watsor_amd.detection.detector import BatchedWorkerMixin, hip_detector_options
        if null_detector:
            HipObjectDetector = NullDetector
        else:
            from watsor_amd.detection.hip_gpu import HipObjectDetector

        class Worker(BatchedWorkerMixin):
            _logger = None

            def _no_frame(self, *a, **k):
                pass

        acct = dict(c_time=0.0, c_calls=0, frames=0)
        shared_fps = fps

        def fps(value=None):                                       # this worker's own share of the (shared) frames/s gauge
            if value is not None:
                acct["frames"] += 1
            return shared_fps(value=value)

        def timed(fn):
            def call(*a):
                t0 = time.perf_counter()
                try:
                    return fn(*a)
                finally:
                    acct["c_time"] += time.perf_counter() - t0
                    acct["c_calls"] += 1
            return call

        bq = shm.BalancedQueueStandIn(q, sems)
        opts = hip_detector_options(frame_buffers, kwargs)
        w = Worker()
        with HipObjectDetector(model_dir, device, opts) as det:
            st = w._hip_state(frame_buffers, det, kwargs)          # binds cameras + frame table (what the first _process does)
            for name in ("submit_bound", "collect_bound", "submit_host", "collect"):
                if hasattr(det, name):
                    setattr(det, name, timed(getattr(det, name)))
            result_q.put(("ready", det.device_name, st["table"] is not None, st["lanes"]))
            t0 = time.perf_counter()
            while not stop_event.is_set():
                w._process(bq, stop_event, frame_buffers, fps, inference_time, det, **kwargs)
            w.drain(fps, inference_time)
            wall = time.perf_counter() - t0
            frames = acct["frames"]
            result_q.put(("done", dict(tag=tag, device=device, wall_s=wall, frames=frames, c_time_s=acct["c_time"], c_calls=acct["c_calls"],
                                       python_us_per_frame=(wall - acct["c_time"]) / max(frames, 1) * 1e6,
                                       inside_library_us_per_frame=acct["c_time"] / max(frames, 1) * 1e6)))
    except Exception:
        result_q.put(("error", traceback.format_exc()))


def run(model_dir, n_cams=8, width=640, height=480, seconds=3.0, costly=True, lanes=4, frame_table=True, producers=2,
        frames_per_buffer=4, max_batch=8, null_detector=False, warm_frames=40, workers=1, gpus=1, check=False, device=0):
    """-> dict(value=frames/s over a `seconds` window of the running worker, p50_ms enqueue -> latch, python_us_per_frame, ...)."""
    import numpy as np
    import shm_standins as shm
    from watsor_amd.synth import synthetic_frame
    ctx = shm.spawn_context()
    samples = ctx.Array("d", 1 + 4096, lock=False)
    if costly:
        cams = {"cam%02d" % c: shm.CostlyFrameBuffer(ctx, frames_per_buffer, width, height, samples=samples) for c in range(n_cams)}
        fps, it = shm.CostlyGauge(ctx), shm.CostlyGauge(ctx, mean=True)
    else:
        cams = {"cam%02d" % c: shm.FrameBuffer(ctx, frames_per_buffer, width, height) for c in range(n_cams)}
        fps, it = shm.Gauge(ctx), shm.Gauge(ctx)
    for c, name in enumerate(sorted(cams)):
        for i, f in enumerate(cams[name].frames):
            np.copyto(np.frombuffer(f.image.get_obj(), np.uint8), synthetic_frame(width, height, 7000 + 16 * c + i).reshape(-1))
    q = ctx.Queue()
    sems = {name: ctx.BoundedSemaphore(1) for name in cams}
    stop, result_q = ctx.Event(), ctx.Queue()
    kwargs = dict(hip_lanes=lanes, hip_frame_table=frame_table, hip_options={"max_batch": max_batch})
    wps = [ctx.Process(target=worker, args=(model_dir, cams, q, sems, stop, fps, it, result_q, kwargs, null_detector, device + k % max(gpus, 1), k))
           for k in range(workers)]
    for wp in wps:
        wp.start()
    procs = []
    try:
        t_give_up = time.time() + 300
        for _ in wps:
            msg = None
            while msg is None:                              # (a worker that died while starting says nothing)
                try:
                    msg = result_q.get(timeout=1)
                except pyqueue.Empty:
                    if not all(wp.is_alive() for wp in wps) or time.time() > t_give_up:
                        raise RuntimeError("a worker process did not come up (exit codes %r)" % ([wp.exitcode for wp in wps],))
            if msg[0] != "ready":
                raise RuntimeError(msg[1])
            _, device, table, eff_lanes = msg
        names = sorted(cams)
        for k in range(producers):
            ready = ctx.Event()
            p = ctx.Process(target=producer, args=(names[k::producers], cams, q, sems, stop, ready))
            p.start()
            procs.append(p)
            ready.wait(120)
        # warm-up: graphs captured, band hints settled
        t_end = time.time() + 60
        while fps.count.value < warm_frames * n_cams and time.time() < t_end:
            time.sleep(0.05)
        samples[0] = 0
        c0, t0 = fps.count.value, time.perf_counter()
        time.sleep(seconds)
        c1, t1 = fps.count.value, time.perf_counter()
        nsamp = int(samples[0])
        lat = np.array(samples[1:1 + min(nsamp, 4096)], dtype=np.float64) * 1e3
    finally:
        stop.set()
    dones = [result_q.get(timeout=120) for _ in wps]
    for p in procs:
        p.join(30)
    for wp in wps:
        wp.join(60)
    for d in dones:
        if d[0] != "done":
            raise RuntimeError(d[1])
    dones.sort(key=lambda d: d[1]["tag"])
    tot = {k: sum(d[1][k] for d in dones) for k in ("frames", "c_time_s", "c_calls")}
    done = ("done", dict(frames=tot["frames"], c_calls=tot["c_calls"],
                         python_us_per_frame=sum(d[1]["python_us_per_frame"] * d[1]["frames"] for d in dones) / max(tot["frames"], 1),
                         inside_library_us_per_frame=sum(d[1]["inside_library_us_per_frame"] * d[1]["frames"] for d in dones) / max(tot["frames"], 1)))
    out = dict(value=round((c1 - c0) / (t1 - t0), 1), unit="frames/s", cameras=n_cams, frame="%dx%d" % (width, height),
               window_s=round(t1 - t0, 3), worker_lanes=eff_lanes, frame_table=bool(table), producers=producers,
               p50_ms_enqueue_to_latch=round(float(np.median(lat)), 3) if lat.size else None,
               p90_ms_enqueue_to_latch=round(float(np.percentile(lat, 90)), 3) if lat.size else None,
               python_us_per_frame=round(done[1]["python_us_per_frame"], 2),
               inside_library_us_per_frame=round(done[1]["inside_library_us_per_frame"], 2),
               library_calls_per_frame=round(done[1]["c_calls"] / max(done[1]["frames"], 1), 3),
               inference_time_gauge_ms=round(it.total.value / max(it.count.value, 1), 4),
               inference_time_observations=int(it.count.value), frames_seen=int(fps.count.value),
               runtime_objects="stand-ins with the reference's locking (tests/shm_standins.py: Costly*)" if costly else "light stand-ins",
               device=device)
    if workers > 1:
        out.update(workers=workers, gpus=gpus, max_batch=max_batch,
                   per_worker=[dict(worker=d[1]["tag"], gpu=d[1]["device"], frames=d[1]["frames"],
                                    frames_per_s=round(d[1]["frames"] / max(d[1]["wall_s"], 1e-9), 1)) for d in dones])
    elif max_batch != 8:
        out["max_batch"] = max_batch
    if check:
        # every dequeued payload latched exactly once and counted once; per camera, how many of its frames went through
        steps = {name: sum(int(f.latch.steps.value) for f in fb.frames) for name, fb in cams.items()}
        out["check"] = dict(latch_steps=sum(steps.values()), fps_calls=int(fps.count.value), worker_frames=tot["frames"],
                            per_camera_steps=[steps[n] for n in sorted(steps)],
                            rows_written=sum(1 for fb in cams.values() for f in fb.frames if f.header.get_obj().detections[0].label >= 1),
                            frames_total=sum(len(fb.frames) for fb in cams.values()))
    if null_detector:
        out["invalid"] = "harness self-test: no detector behind the worker"
    return out


if __name__ == "__main__":
    from watsor_amd import engine
    from watsor_amd.synth import synthetic_weights
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("cameras", nargs="?", type=int, default=8)
    ap.add_argument("seconds", nargs="?", type=float, default=3.0)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--max-batch", type=int, default=8)
    a = ap.parse_args()
    d = "/tmp/wz_worker_bench_%d" % os.getpid()
    os.makedirs(d, exist_ok=True)
    engine.save_engine(engine.build_engine(synthetic_weights(1234)), os.path.join(d, "mi355x.bin"))
    for table in ((True, False) if a.workers == 1 else (True,)):
        print(json.dumps(run(d, a.cameras, seconds=a.seconds, frame_table=table, workers=a.workers, gpus=a.gpus,
                             max_batch=a.max_batch)), flush=True)
