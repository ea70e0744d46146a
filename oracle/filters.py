"""Oracle (test infrastructure): the per-camera filters that run right behind the detector.

`ConfidenceFilter` / `AreaFilter` restate `watsor/filter/confidence.py:10-19` and
`watsor/filter/area.py:10-26` line by line (tests/golden/make_filter_golden.py pins them against the
reference's own classes).  `MaskFilter` restates `watsor/filter/mask.py:17-59` *literally* -- zones are
contour polygons, a hit is "closed box polygon intersects closed zone polygon" -- with exact integer
geometry standing in for shapely and `oracle/zones.py` standing in for OpenCV's contour finder; the
HIP path uses the raster reformulation and is checked against this.

Pinned by the reference's known-answer tests `watsor/test/test_filter.py:14-74`.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import zones as oz

COCO_CLASSES = None


def coco_classes() -> List[str]:
    """Label table (watsor/config/coco.py:14-106); the product's copy is checked against the reference's."""
    global COCO_CLASSES
    if COCO_CLASSES is None:
        from watsor_amd.coco import COCO_CLASSES as c
        COCO_CLASSES = c
    return COCO_CLASSES


class ConfidenceFilter(object):
    def __init__(self, camera_config):
        self.indexes: Dict[int, float] = {}
        for entry in camera_config['detect']:
            coco_class = next(iter(entry))
            idx = coco_classes().index(coco_class)
            self.indexes[idx] = entry[coco_class]['confidence'] / 100           # confidence.py:15

    def __call__(self, detection) -> bool:
        confidence = self.indexes.get(detection.label, None)
        return confidence is not None and detection.confidence >= confidence    # confidence.py:17-19


class AreaFilter(object):
    def __init__(self, camera_config):
        self.indexes: Dict[int, float] = {}
        for entry in camera_config['detect']:
            coco_class = next(iter(entry))
            idx = coco_classes().index(coco_class)
            width, height = camera_config['width'], camera_config['height']
            max_area = self.area_xyxy(0, 0, width - 1, height - 1)              # area.py:16
            self.indexes[idx] = entry[coco_class]['area'] / 100 * max_area       # area.py:17

    def __call__(self, detection) -> bool:
        area = self.indexes.get(detection.label, None)
        bb = detection.bounding_box
        return area is not None and self.area_xyxy(bb.x_min, bb.y_min, bb.x_max, bb.y_max) >= area

    @staticmethod
    def area_xyxy(x_min, y_min, x_max, y_max):
        return abs((x_max - x_min + 1) * (y_max - y_min + 1))                   # area.py:24-26


class MaskFilter(object):
    """mask.py:17-59 on an alpha plane (H,W uint8) instead of a PNG path."""

    def __init__(self, camera_config, alpha: Optional[np.ndarray] = None):
        if alpha is None:
            alpha = oz.read_alpha(camera_config['mask'], camera_config.get('width'), camera_config.get('height'))
        self.polygons = oz.zone_polygons(alpha)                                 # sorted contours (mask.py:22-26)
        self.polygons_by_zone: Dict[int, List] = {}
        for entry in camera_config['detect']:
            coco_class = next(iter(entry))
            index = coco_classes().index(coco_class)
            zones = entry[coco_class]['zones']
            if len(zones) == 0:
                continue
            for z in zones:
                assert 0 < z <= len(self.polygons), \
                    "There is no zone {} in mask {}".format(z, camera_config.get('mask'))
            self.polygons_by_zone[index] = [p if idx + 1 in zones else None for idx, p in enumerate(self.polygons)]

    def __call__(self, detection) -> bool:
        bb = detection.bounding_box
        polygons = self.polygons_by_zone.get(detection.label, self.polygons)
        result = False
        z = 0
        p = 0
        while p < len(polygons) and z < len(detection.zones):
            if polygons[p] is not None and oz.box_intersects_polygon(bb.x_min, bb.y_min, bb.x_max, bb.y_max,
                                                                      polygons[p]):
                detection.zones[z] = p + 1
                z += 1
                result = True
            p += 1
        return result


def apply_filters(filters: Sequence, detections) -> np.ndarray:
    """`label > 0 and all(f(d) for f in filters)` per row (watsor/filter/track.py:26), short-circuit kept."""
    out = np.zeros(len(detections), np.uint8)
    for i, d in enumerate(detections):
        out[i] = 1 if (d.label > 0 and all(f(d) for f in filters)) else 0
    return out
