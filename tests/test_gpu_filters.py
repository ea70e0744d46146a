"""GPU confidence / area / zone filters vs the oracle's literal (polygon) filters: bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import make_engine
from oracle import filters as of
from watsor_amd.filter.hip_filter import HipCameraFilter
from watsor_amd.runtime import ROW_DTYPE
from watsor_amd.share import BoundingBox, Detection
from watsor_amd.synth import synthetic_frame

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def eng(model_dir):
    e = make_engine(model_dir, max_batch=4)
    yield e
    e.close()


def porch_alpha():
    z = np.load(os.path.join(GOLDEN, "porch_zones.npz"))
    h, w = z["shape"]
    opaque = np.unpackbits(z["opaque"])[:h * w].reshape(h, w).astype(bool)
    return np.where(opaque, 255, 204).astype(np.uint8)


def golden_config():
    return json.load(open(os.path.join(GOLDEN, "filters.json")))


def rows_from(dets):
    rows = np.zeros(100, ROW_DTYPE)
    for i, (label, conf, box) in enumerate(dets):
        rows[i]["label"], rows[i]["confidence"] = label, conf
        rows[i]["x_min"], rows[i]["y_min"], rows[i]["x_max"], rows[i]["y_max"] = box
    return rows


def oracle_verdict(filters, rows):
    """Literal reference semantics on a copy of the rows: returns (pass[100], zones[100,10])."""
    out_pass = np.zeros(100, np.uint8)
    zones = np.zeros((100, 10), np.int32)
    for i in range(100):
        d = Detection(label=int(rows[i]["label"]), confidence=float(rows[i]["confidence"]),
                      bounding_box=BoundingBox(int(rows[i]["x_min"]), int(rows[i]["y_min"]), int(rows[i]["x_max"]),
                                               int(rows[i]["y_max"])))
        out_pass[i] = 1 if (d.label > 0 and all(f(d) for f in filters)) else 0
        zones[i] = list(d.zones)
    return out_pass, zones


def test_confidence_area_reference_fixture(eng):
    g = golden_config()
    flt = HipCameraFilter(eng, 3, g["config"])
    rows_all = g["rows"]
    for start in range(0, len(rows_all), 100):
        chunk = rows_all[start:start + 100]
        rows = rows_from([(r["label"], r["confidence"], r["box"]) for r in chunk])
        got = flt.filter_rows(rows)
        want = [1 if (r["label"] > 0 and r["conf_pass"] and r["area_pass"]) else 0 for r in chunk]
        assert got[:len(chunk)].tolist() == want
        assert not rows["zones"].any()
    flt.close()


@pytest.mark.parametrize("seed", range(3))
def test_mask_filter_matches_polygon_oracle(eng, seed):
    cfg = dict(golden_config()["config"])
    cfg["detect"] = [{"person": {"area": 2, "confidence": 30, "zones": []}},
                     {"car": {"area": 1, "confidence": 20, "zones": [2]}},
                     {"truck": {"area": 10, "confidence": 50, "zones": [1, 2]}}]
    alpha = porch_alpha()
    flt = HipCameraFilter(eng, 7, cfg, alpha=alpha)
    assert flt.num_zones == 2
    filters = [of.ConfidenceFilter(cfg), of.AreaFilter(cfg), of.MaskFilter(cfg, alpha=alpha)]
    rng = np.random.default_rng(seed)
    dets = []
    for i in range(100):
        x0, y0 = int(rng.integers(0, 630)), int(rng.integers(0, 470))
        x1, y1 = int(rng.integers(x0, 640)), int(rng.integers(y0, 480))
        if i % 9 == 0:
            x1, y1 = x0, y0                                            # single-pixel boxes
        dets.append((int(rng.choice([0, 1, 3, 8, 2])), float(np.float32(rng.random())), (x0, y0, x1, y1)))
    rows = rows_from(dets)
    want_pass, want_zones = oracle_verdict(filters, rows.copy())
    got_pass = flt.filter_rows(rows)
    np.testing.assert_array_equal(got_pass, want_pass)
    np.testing.assert_array_equal(rows["zones"], want_zones)
    assert want_pass.sum() > 5 and (want_zones > 0).sum() > 5
    flt.close()


def test_kat_mask_on_gpu(eng):
    """watsor/test/test_filter.py:51-74: right half opaque; (20,20,40,80) misses, (20,20,80,80) hits zone 1."""
    alpha = np.zeros((100, 100), np.uint8)
    alpha[:, 50:] = 255
    cfg = {"width": 100, "height": 100, "detect": [{"person": {"area": 0, "confidence": 0, "zones": []}}]}
    flt = HipCameraFilter(eng, 1, cfg, alpha=alpha)
    rows = rows_from([(1, 0.70, (20, 20, 40, 80)), (1, 0.70, (20, 20, 80, 80))])
    got = flt.filter_rows(rows)
    assert got[0] == 0 and got[1] == 1 and rows["zones"][1][0] == 1 and not rows["zones"][0].any()
    flt.close()


def test_filters_inside_detect_batch(eng):
    """Rows of frames tagged with a camera id come back filtered; untagged frames are untouched."""
    cfg = dict(golden_config()["config"])
    cfg["detect"] = [{name: {"area": 1, "confidence": 10, "zones": []}}
                     for name in ("person", "car", "bench", "bird", "cat", "dog")]
    alpha = porch_alpha()
    flt = HipCameraFilter(eng, 5, cfg, alpha=alpha)
    filters = [of.ConfidenceFilter(cfg), of.AreaFilter(cfg), of.MaskFilter(cfg, alpha=alpha)]
    frames = [synthetic_frame(640, 480, 11), synthetic_frame(640, 480, 12)]
    rows = [np.zeros(100, ROW_DTYPE) for _ in frames]
    passes = [np.full(100, 9, np.uint8) for _ in frames]
    eng.detect_batch(frames, rows, cams=[5, -1], out_pass=passes)
    plain = rows[0].copy()
    plain["zones"] = 0
    want_pass, want_zones = oracle_verdict(filters, plain)
    np.testing.assert_array_equal(passes[0], want_pass)
    np.testing.assert_array_equal(rows[0]["zones"], want_zones)
    assert not rows[1]["zones"].any()
    np.testing.assert_array_equal(passes[1], (rows[1]["label"] > 0).astype(np.uint8))
    with pytest.raises(ValueError):                                  # camera filter was set for 640x480
        eng.detect_batch([synthetic_frame(1280, 720, 1)], rows[:1], cams=[5])
    flt.close()
