"""Oracle (test infrastructure): the reference CPU plugin's `detect()` restated end to end.

Follows `watsor/detection/tensorflow_cpu.py:74-92` line by line for the part that *is* reference
code (timing of the detect function only, the 100-iteration row fill, truncation toward zero,
no clamp, all rows written including the zero-score padding) and stands in for `sess.run`
(`:94-121`) with the restated graph of `oracle/{preprocess,ssd_mobilenet_v2,postprocess}.py`.

`OracleObjectDetector` satisfies the reference's duck-typed plugin protocol:
`__init__(model_path)`, `device_name`, `__enter__/__exit__`, `detect(image_shape, image_np,
detections) -> ms`, so it can run under the *unmodified* `ObjectDetector` worker
(`watsor/detection/detector.py:84-112`) -- BASELINE config 1 -- and serves as `bench.py`'s
`cpu_baseline` (kind "port").
"""
from __future__ import annotations

from time import time
from typing import Dict, Optional

import numpy as np

from . import postprocess as post
from . import preprocess as pre
from .ssd_mobilenet_v2 import OracleNet


def fill_rows(image_shape, boxes: np.ndarray, label_codes: np.ndarray, scores: np.ndarray, detections) -> None:
    """tensorflow_cpu.py:79-90.  `boxes` float32 [N,4] (ymin,xmin,ymax,xmax), normalised.

    The reference image pins numpy 1.23 (`docker/Dockerfile.base:33`) where `np.float32 * int`
    promotes to float64, i.e. the product is exact in double before `int()` truncates toward
    zero; that exact-double form is what is restated here.
    """
    d = 0
    max_width = image_shape[1] - 1
    max_height = image_shape[0] - 1
    while d < len(scores) and d < len(detections):
        detection = detections[d]
        detection.label = int(label_codes[d])
        detection.confidence = float(scores[d])
        detection.bounding_box.y_min = int(float(boxes[d][0]) * max_height)
        detection.bounding_box.x_min = int(float(boxes[d][1]) * max_width)
        detection.bounding_box.y_max = int(float(boxes[d][2]) * max_height)
        detection.bounding_box.x_max = int(float(boxes[d][3]) * max_width)
        d += 1


def rows_as_array(image_shape, boxes, label_codes, scores) -> Dict[str, np.ndarray]:
    """Vectorised fill_rows for bulk comparisons: label i32, confidence f64, box i32 [N,4] (x_min,y_min,x_max,y_max)."""
    mh, mw = float(image_shape[0] - 1), float(image_shape[1] - 1)
    b = boxes.astype(np.float64)
    box = np.stack([np.trunc(b[:, 1] * mw), np.trunc(b[:, 0] * mh), np.trunc(b[:, 3] * mw), np.trunc(b[:, 2] * mh)], 1)
    return {"label": label_codes.astype(np.int32), "confidence": scores.astype(np.float64),
            "box": box.astype(np.int32)}


class OracleObjectDetector:
    """CPU restatement of `TensorFlowObjectDetector` (tensorflow_cpu.py:10-121) on explicit weights."""

    def __init__(self, model_path=None, weights: Optional[Dict[str, np.ndarray]] = None, size: int = 300,
                 fast_post: bool = False, half_pixel_centers: bool = False, clip_after_nms: bool = False, post_config: Optional[dict] = None):
        """fast_post: post-process in one global score order (`postprocess.multiclass_nms_global_order`, held equal to the
        literal class-by-class version by tests) -- for the CPU baseline of bench.py; the checker uses the literal one."""
        self._fast_post = fast_post
        # the two steps that differ between exporter generations (oracle/preprocess.py, oracle/postprocess.py) and the
        # post-processing constants a graph may carry (score_thr, iou_thr, max_per_class, max_total)
        self._half_pixel, self._clip_after, self._post = half_pixel_centers, clip_after_nms, dict(post_config or {})
        if weights is None:
            import os
            path = os.path.join(model_path, "oracle.npz")
            if not os.path.isfile(path):
                raise FileNotFoundError(path)      # detector.py:97-98 logs this and the worker exits
            with np.load(path) as z:
                weights = {k: z[k] for k in z.files}
        self._net = OracleNet(weights)
        self._size = size
        self._anchors = post.anchors_center_size(post.generate_anchors(size))

    @property
    def device_name(self):
        return "CPU"

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        pass

    def raw(self, image_np: np.ndarray):
        """(boxes[100,4], classes[100] 1-based float, scores[100], box_enc, logits) for one frame."""
        x = pre.preprocess(image_np, self._size, self._half_pixel)[None]
        be, cl, _ = self._net.forward(x)
        b, s, c, _ = post.postprocess(be[0], cl[0], self._anchors, fast=self._fast_post and not self._clip_after,
                                      clip_after_nms=self._clip_after, **self._post)
        return b, c, s, be[0], cl[0]

    def detect(self, image_shape, image_np, detections) -> float:
        inference_start_time = time()
        boxes, label_codes, scores, _, _ = self.raw(image_np)
        inference_time = (time() - inference_start_time) * 1000
        fill_rows(image_shape, boxes, label_codes, scores, detections)
        return inference_time
