"""The `worker_spawned` harness of bench.py (tools/worker_bench.py) without a GPU: spawned worker + spawned producers over one
real multiprocessing.Queue, shared-memory frame buffers with the reference's locking, a detector that detects nothing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_worker_harness_counts_frames_and_latencies():
    import worker_bench
    r = worker_bench.run("/nonexistent", n_cams=4, width=64, height=48, seconds=0.6, null_detector=True, warm_frames=10)
    assert r["invalid"] and r["frame_table"] and r["worker_lanes"] == 4
    assert r["value"] > 200 and r["frames_seen"] > 100
    assert r["p50_ms_enqueue_to_latch"] is not None and 0 < r["p50_ms_enqueue_to_latch"] < 500
    assert 0 < r["python_us_per_frame"] < 5000 and r["library_calls_per_frame"] <= 2.0
    assert 1 <= r["inference_time_observations"] <= r["frames_seen"]


def test_two_worker_processes_on_one_queue_latch_every_payload_once():
    """VERDICT r3 next #5b/c, harness side: two spawned workers drain ONE multiprocessing.Queue under the one-queued-frame-per-camera
    rule; every dequeued payload is latched exactly once and counted once, both workers get work, no camera is left out."""
    import worker_bench
    r = worker_bench.run("/nonexistent", n_cams=6, width=64, height=48, seconds=0.8, null_detector=True, warm_frames=10,
                         workers=2, gpus=1, costly=False, check=True, max_batch=4)
    assert r["workers"] == 2 and len(r["per_worker"]) == 2 and all(w["frames"] > 50 for w in r["per_worker"])
    c = r["check"]
    assert c["latch_steps"] == c["fps_calls"] == c["worker_frames"] > 200
    per_cam = c["per_camera_steps"]
    # ("no camera is left out", not "all cameras alike": the two producer processes feed cameras 0, 2, 4 and 1, 3, 5 and run at whatever rates a busy host
    #  gives them -- [1379, 917, 1357, 907, 1337, 899] has been seen on this container)
    assert min(per_cam) > 0 and max(per_cam) <= 3.0 * min(per_cam) + 50, per_cam
