"""Child-process side of tests/test_gpu_worker.py (module level: the spawn start method imports it by name)."""
import queue
import traceback


def run_worker(model_dir, frame_buffers, payload_batches, fps, inference_time, camera_configs, drop, result_queue,
               asynchronous=True, hip_options=None):
    """What `ObjectDetector._run` does (`watsor/detection/detector.py:84-100`) around `BatchedWorkerMixin._process`:
    construct the plugin IN THIS PROCESS (HIP context after spawn), spin over a queue of payloads, exit."""
    try:
        from watsor_amd.detection.detector import BatchedWorkerMixin, hip_detector_options
        from watsor_amd.detection.hip_gpu import HipObjectDetector

        class Worker(BatchedWorkerMixin):
            _logger = None

            def _no_frame(self, *a, **k):
                pass

        q = queue.Queue()
        kwargs = dict(hip_cameras=camera_configs, hip_drop=drop, hip_async=asynchronous, hip_lanes=2, hip_options=hip_options)
        opts = hip_detector_options(frame_buffers, kwargs)
        w = Worker()
        with HipObjectDetector(model_dir, 0, opts) as det:
            for batch in payload_batches:
                for p in batch:
                    q.put(p)
                # one _process call drains what is queued (up to max_batch) -- like the worker's spin loop does
                while not q.empty():
                    w._process(q, None, frame_buffers, fps, inference_time, det, **kwargs)
            w.drain(fps, inference_time)
            result_queue.put(("ok", det.device_name, dict(opts), len(getattr(det, "_HipObjectDetector__pinned"))))
    except Exception:
        result_queue.put(("error", traceback.format_exc(), None, 0))
