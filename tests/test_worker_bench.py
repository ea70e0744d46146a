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
