"""Per-camera Confidence / Area / Mask filters evaluated on the GPU.

Host-side construction mirrors the three reference constructors so the same (normalised) camera
config dictionary can be handed over unchanged:

  * `ConfidenceFilter.__init__` (`watsor/filter/confidence.py:10-15`): threshold = confidence / 100
  * `AreaFilter.__init__`       (`watsor/filter/area.py:10-17`):  threshold = area / 100 * W*H
  * `MaskFilter.__init__`       (`watsor/filter/mask.py:17-42`):   zones from the mask's alpha plane,
    per-label allow-list from `zones`, same assertion messages (`mask.py:62-75,36-37`)

The per-detection part (`__call__` of the three classes, combined by `all()` in
`watsor/filter/track.py:26`) runs in `wz_k_rows` / `wz_k_filter_rows` (csrc/k_post.hip): the rows of
every frame tagged with this camera id come back with `zones[]` filled in and a pass byte per row.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from ..coco import COCO_CLASSES
from ..runtime import HipEngine, zones_from_alpha

NUM_LABELS = len(COCO_CLASSES)


def read_mask_alpha(filename, width=None, height=None) -> np.ndarray:
    """`get_alpha_channel` of mask.py:62-75 (PIL instead of cv2.imread, same assertions)."""
    from PIL import Image
    try:
        image = Image.open(filename)
        image.load()
    except Exception:
        image = None
    assert image is not None, "Error reading mask file {}".format(filename)
    mask_image = np.array(image)
    assert len(mask_image.shape) == 3 and mask_image.shape[2] == 4, \
        "Mask image {} is not of 32 bit color".format(filename)
    if width is not None and height is not None:
        assert mask_image.shape[0] == height and mask_image.shape[1] == width, \
            "The size of mask image {} doesn't match {}x{}".format(filename, width, height)
    return np.ascontiguousarray(mask_image[:, :, 3])


class HipCameraFilter:
    def __init__(self, engine: HipEngine, cam: int, camera_config: dict, alpha: Optional[np.ndarray] = None,
                 drop: bool = False):
        """drop=True: rows failing the filters leave the GPU as all-zero rows (`wz_set_camera_drop`), which is what
        a sieve with a filter-less `HipTrackFilter()` expects."""
        self.engine, self.cam = engine, cam
        width, height = camera_config['width'], camera_config['height']
        conf = np.full(NUM_LABELS, math.nan)
        area = np.full(NUM_LABELS, math.nan)
        max_area = abs(((width - 1) - 0 + 1) * ((height - 1) - 0 + 1))          # area.py:16,24-26
        for entry in camera_config['detect']:
            coco_class = next(iter(entry))
            idx = COCO_CLASSES.index(coco_class)
            conf[idx] = entry[coco_class]['confidence'] / 100
            area[idx] = entry[coco_class]['area'] / 100 * max_area
        fill = allow = None
        self.centroids = np.zeros((0, 2), np.int32)
        if alpha is None and 'mask' in camera_config:
            alpha = read_mask_alpha(camera_config['mask'], width, height)
        if alpha is not None:
            fill, self.centroids = zones_from_alpha(alpha)
            nz = fill.shape[0]
            allow = np.ones((NUM_LABELS, nz), np.uint8)
            for entry in camera_config['detect']:
                coco_class = next(iter(entry))
                idx = COCO_CLASSES.index(coco_class)
                zones = entry[coco_class]['zones']
                if len(zones) == 0:
                    continue
                for z in zones:
                    assert 0 < z <= nz, "There is no zone {} in mask {}".format(z, camera_config.get('mask'))
                allow[idx] = [1 if i + 1 in zones else 0 for i in range(nz)]
        self.num_zones = 0 if fill is None else fill.shape[0]
        engine.set_camera_filter(cam, width, height, conf, area, fill, allow)
        if drop:
            engine.set_camera_drop(cam, True)

    def filter_rows(self, rows) -> np.ndarray:
        """Runs the camera's filters over 100 rows in place (zones written); returns pass[100]."""
        return self.engine.filter_rows(self.cam, rows)

    def close(self):
        self.engine.clear_camera_filter(self.cam)
