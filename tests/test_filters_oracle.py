"""Filter oracle + host-side zone extraction, pinned by the reference's known-answer tests
(watsor/test/test_filter.py:14-74), by fixtures generated from the reference's own classes
(tests/golden/make_filter_golden.py) and by the zone statistics of the reference's porch.png."""
import json
import os

import numpy as np
import pytest
from PIL import Image, ImageDraw

from oracle import filters as of
from oracle import zones as oz
from watsor_amd.runtime import zones_from_alpha
from watsor_amd.share import BoundingBox, Detection

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def det(label, confidence, box=(0, 0, 0, 0)):
    return Detection(label=label, confidence=confidence, bounding_box=BoundingBox(*box))


# ---- known-answer tests restated from watsor/test/test_filter.py -----------------------------------
def test_confidence_kat():                      # test_filter.py:14-22
    f = of.ConfidenceFilter({'detect': [{'person': {'confidence': 50}}]})
    assert f(det(1, 0.70)) and not f(det(1, 0.40)) and not f(det(2, 0.70))


def test_area_kat():                            # test_filter.py:24-36
    f = of.AreaFilter({'width': 100, 'height': 100, 'detect': [{'person': {'area': 50}}]})
    assert f(det(1, 0.7, (0, 0, 100, 50)))
    assert not f(det(1, 0.7, (0, 0, 50, 50)))
    assert not f(det(2, 0.7, (0, 0, 100, 50)))


def kat_alpha():
    """test_filter.py:51-55: 100x100 RGBA, alpha = rectangle (50,0)-(100,100) filled with 255."""
    with Image.new("L", (100, 100)) as alpha:
        ImageDraw.Draw(alpha).rectangle((50, 0, alpha.width, alpha.height), fill=255)
        return np.array(alpha)


def test_mask_kat(tmp_path):                    # test_filter.py:38-74
    with pytest.raises(AssertionError, match="Error reading mask file"):
        of.MaskFilter({'width': 1, 'height': 1, 'mask': 'notafile.png', 'detect': []})
    p = str(tmp_path / "m.png")
    Image.new('RGB', (10, 10)).save(p)
    with pytest.raises(AssertionError, match="Mask image .+ is not of 32 bit color"):
        of.MaskFilter({'width': 10, 'height': 10, 'mask': p, 'detect': []})
    img = Image.new('RGBA', (100, 100))
    img.putalpha(Image.fromarray(kat_alpha()))
    img.save(p)
    with pytest.raises(AssertionError, match="The size of mask image .+ doesn't match"):
        of.MaskFilter({'width': 50, 'height': 50, 'mask': p, 'detect': []})
    f = of.MaskFilter({'width': 100, 'height': 100, 'mask': p, 'detect': []})
    assert not f(det(1, 0.7, (20, 20, 40, 80)))
    d = det(1, 0.7, (20, 20, 80, 80))
    assert f(d) and d.zones[0] == 1


def test_product_mask_reader_raises_like_the_reference(tmp_path):
    from watsor_amd.filter.hip_filter import read_mask_alpha
    with pytest.raises(AssertionError, match="Error reading mask file"):
        read_mask_alpha('notafile.png', 1, 1)
    p = str(tmp_path / "m.png")
    Image.new('RGB', (10, 10)).save(p)
    with pytest.raises(AssertionError, match="Mask image .+ is not of 32 bit color"):
        read_mask_alpha(p, 10, 10)
    Image.new('RGBA', (100, 100)).save(p)
    with pytest.raises(AssertionError, match="The size of mask image .+ doesn't match"):
        read_mask_alpha(p, 50, 50)


# ---- fixtures generated from the reference's own ConfidenceFilter / AreaFilter ------------------------
def test_confidence_area_match_reference_fixture():
    g = json.load(open(os.path.join(GOLDEN, "filters.json")))
    cf, af = of.ConfidenceFilter(g["config"]), of.AreaFilter(g["config"])
    assert len(g["rows"]) == 400
    for r in g["rows"]:
        d = det(r["label"], r["confidence"], r["box"])
        assert cf(d) == r["conf_pass"] and af(d) == r["area_pass"], r
    assert 30 < sum(r["conf_pass"] for r in g["rows"]) < 370


@pytest.mark.reference
def test_coco_table_and_filters_against_reference(reference_on_path):
    from watsor.config.coco import COCO_CLASSES as ref
    from watsor.filter.area import AreaFilter
    from watsor.filter.confidence import ConfidenceFilter
    from watsor_amd.coco import COCO_CLASSES
    assert list(ref) == list(COCO_CLASSES)
    cfg = json.load(open(os.path.join(GOLDEN, "filters.json")))["config"]
    rng = np.random.default_rng(5)
    a, b = ConfidenceFilter(cfg), of.ConfidenceFilter(cfg)
    c, d = AreaFilter(cfg), of.AreaFilter(cfg)
    for _ in range(300):
        x0, y0 = int(rng.integers(0, 600)), int(rng.integers(0, 440))
        dd = det(int(rng.choice([0, 1, 3, 8, 5])), float(np.float32(rng.random())),
                 (x0, y0, int(rng.integers(x0, 640)), int(rng.integers(y0, 480))))
        assert a(dd) == b(dd) and c(dd) == d(dd)


# ---- zones: host extraction (C++, raster) vs oracle (Python, polygons) -------------------------------
def porch_opaque():
    z = np.load(os.path.join(GOLDEN, "porch_zones.npz"))
    h, w = z["shape"]
    return np.unpackbits(z["opaque"])[:h * w].reshape(h, w).astype(bool), z["zone_pixels"]


def test_porch_zones_match_survey_statistics():
    opaque, pixels = porch_opaque()
    alpha = np.where(opaque, 255, 204).astype(np.uint8)
    fill, cent = zones_from_alpha(alpha)
    assert fill.shape == (2, 480, 640)
    assert fill.reshape(2, -1).sum(1).tolist() == pixels.tolist()          # 42 853 and 58 038 px
    assert abs(cent[0][0] - 132) <= 1 and abs(cent[0][1] - 377) <= 1      # centroids ~(132,377), ~(424,370)
    assert abs(cent[1][0] - 424) <= 1 and abs(cent[1][1] - 370) <= 1
    polys = oz.zone_polygons(alpha)
    assert len(polys) == 2
    for z, poly in enumerate(polys):
        m00, m10, m01 = oz.contour_moments(poly)
        assert (int(m10 / m00), int(m01 / m00)) == tuple(cent[z])


def random_blobs(rng, h, w, n):
    alpha = rng.integers(0, 255, (h, w)).astype(np.uint8)                  # anything but 255
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cy, cx = rng.integers(8, h - 8), rng.integers(8, w - 8)
        ry, rx = rng.integers(4, h // 4), rng.integers(4, w // 4)
        if rng.random() < 0.5:
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
        else:
            m = (np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)
        alpha[m] = 255
        if rng.random() < 0.5:                                             # punch a hole, maybe with an island
            hm = ((yy - cy) / max(ry // 2, 1)) ** 2 + ((xx - cx) / max(rx // 2, 1)) ** 2 <= 1
            alpha[hm] = 254
            if rng.random() < 0.5:
                alpha[cy, cx] = 255
    return alpha


@pytest.mark.parametrize("seed", range(6))
def test_zone_fill_equals_lattice_points_of_contour_polygons(seed):
    rng = np.random.default_rng(seed)
    alpha = random_blobs(rng, 60, 80, 3)
    fill, cent = zones_from_alpha(alpha)
    ofill = oz.zone_fill(alpha)                       # polygon interior by exact point-in-polygon
    assert fill.shape == ofill.shape
    np.testing.assert_array_equal(fill, ofill)
    for z, poly in enumerate(oz.zone_polygons(alpha)):
        m00, m10, m01 = oz.contour_moments(poly)
        assert (int(m10 / m00), int(m01 / m00)) == tuple(cent[z])


@pytest.mark.parametrize("seed", range(4))
def test_lattice_rule_equals_polygon_intersection(seed):
    """box polygon intersects zone polygon  <=>  a filled-zone pixel lies in the closed box."""
    rng = np.random.default_rng(100 + seed)
    alpha = random_blobs(rng, 48, 64, 3)
    fill, _ = zones_from_alpha(alpha)
    polys = oz.zone_polygons(alpha)
    for _ in range(400):
        x0, y0 = int(rng.integers(0, 64)), int(rng.integers(0, 48))
        x1, y1 = int(rng.integers(x0, 64)), int(rng.integers(y0, 48))
        if rng.random() < 0.15:
            x1 = x0                                   # degenerate boxes follow the lattice rule too
        for z, poly in enumerate(polys):
            assert oz.box_intersects_polygon(x0, y0, x1, y1, poly) == bool(fill[z, y0:y1 + 1, x0:x1 + 1].any())


def test_thin_zone_raises_zero_division_like_the_reference():
    alpha = np.zeros((20, 20), np.uint8)
    alpha[5, 3:12] = 255                               # 1-px line: contour polygon has zero area
    with pytest.raises(ZeroDivisionError):
        zones_from_alpha(alpha)
    with pytest.raises(ZeroDivisionError):
        oz.zone_polygons(alpha)
