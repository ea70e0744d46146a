"""Generates tests/golden/filters.json by running the REFERENCE's own filter classes
(`watsor/filter/confidence.py`, `watsor/filter/area.py`, imported from /root/reference) on seeded
detections, and tests/golden/porch_zones.npz from the reference's `config/porch.png`.

Run in the build container only:   python tests/golden/make_filter_golden.py
The fixtures travel to the GPU box; /root/reference does not.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from watsor.filter.area import AreaFilter                   # noqa: E402  (reference)
from watsor.filter.confidence import ConfidenceFilter       # noqa: E402  (reference)
from watsor.stream.share import BoundingBox, Detection      # noqa: E402  (reference)

# detect section of the reference's config/config.yaml:69-79 after normalisation
# (defaults area 10, confidence 50, zones [] -- watsor/config/schema.py:87-98)
CONFIG = {
    "width": 640, "height": 480,
    "detect": [
        {"person": {"area": 20, "confidence": 60, "zones": []}},
        {"car": {"area": 10, "confidence": 50, "zones": [1, 3, 5]}},
        {"truck": {"area": 10, "confidence": 50, "zones": []}},
    ],
}


def main():
    rng = np.random.Generator(np.random.PCG64(2024))
    conf_f, area_f = ConfidenceFilter(CONFIG), AreaFilter(CONFIG)
    rows = []
    labels = [0, 1, 3, 8, 2, 14, 90]
    for i in range(400):
        label = int(rng.choice(labels))
        # float32 scores widened to double, like the detector writes them; include exact thresholds
        score = float(np.float32(rng.choice([rng.random(), 0.5, 0.6, np.nextafter(np.float32(0.6), np.float32(0))])))
        x0, y0 = int(rng.integers(0, 600)), int(rng.integers(0, 440))
        x1, y1 = int(rng.integers(x0, 640)), int(rng.integers(y0, 480))
        if i % 37 == 0:      # boxes whose area sits exactly on / next to the threshold (20% resp. 10% of 640*480)
            x0, y0, x1, y1 = 0, 0, 255, 239 + (i // 37) % 3 - 1
        d = Detection(label=label, confidence=score, bounding_box=BoundingBox(x0, y0, x1, y1))
        rows.append(dict(label=label, confidence=score, box=[x0, y0, x1, y1],
                         conf_pass=bool(conf_f(d)), area_pass=bool(area_f(d))))
    json.dump(dict(config=CONFIG, rows=rows), open(os.path.join(HERE, "filters.json"), "w"))

    from PIL import Image
    alpha = np.array(Image.open("/root/reference/config/porch.png"))[:, :, 3]
    np.savez_compressed(os.path.join(HERE, "porch_zones.npz"),
                        opaque=np.packbits(alpha == 255),      # the only property of alpha the reference uses
                        shape=np.array(alpha.shape),
                        # SURVEY.md §4 (probed with the reference's file): two alpha==255 zones
                        zone_pixels=np.array([42853, 58038]))
    print("wrote filters.json (%d rows), porch_zones.npz" % len(rows))


if __name__ == "__main__":
    main()
