"""GPU confidence / area / zone filters vs the oracle's literal (polygon) filters: bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import make_engine
from oracle import filters as of
from oracle import zones as oz
from watsor_amd.coco import COCO_CLASSES
from watsor_amd.filter.hip_filter import HipCameraFilter
from watsor_amd.runtime import ROW_DTYPE
from watsor_amd.share import BoundingBox, Detection
from watsor_amd.synth import synthetic_frame, synthetic_zone_mask as blob_mask

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


def test_many_cameras_with_masks_and_thresholds_mixed_resolutions(model_dir):
    """BASELINE configs[3]/[4] in small: 1920x1080 and 640x480 cameras, each with its own zone mask and the
    thresholds of the reference's sample config (config/config.yaml:69-79), batched together.  The rows come
    from the GPU detector; pass bytes and zones[] must equal the oracle's literal filters on those rows."""
    e = make_engine(model_dir, max_batch=8)
    try:
        cams, flts = [], []
        for cam in range(6):
            w, h = (1920, 1080) if cam % 2 == 0 else (640, 480)
            alpha = blob_mask(w, h, 100 + cam, 2 + cam % 5)
            nz = len(oz.zone_polygons(alpha))                          # zones a config names must exist (mask.py:36-37)
            cfg = {"width": w, "height": h,
                   "detect": [{"person": {"area": 20, "confidence": 60, "zones": []}},
                              {"car": {"area": 10, "confidence": 50, "zones": [z for z in (1, 3, 5) if z <= nz]}},
                              {"truck": {"area": 10, "confidence": 50, "zones": []}},
                              {"bench": {"area": 1, "confidence": 30, "zones": [min(2, nz)]}},
                              # the seeded random-init network mostly reports label 14: let it through on odd cameras
                              {COCO_CLASSES[14]: {"area": 1, "confidence": 30, "zones": [] if cam % 2 else [1]}}]}
            flt = HipCameraFilter(e, 20 + cam, cfg, alpha=alpha)
            assert flt.num_zones >= 1
            cams.append((20 + cam, w, h, [of.ConfidenceFilter(cfg), of.AreaFilter(cfg), of.MaskFilter(cfg, alpha=alpha)]))
            flts.append(flt)
        order = [0, 1, 2, 3, 4, 5, 0, 3]                               # 8 frames, cameras repeated
        frames = [synthetic_frame(cams[c][1], cams[c][2], 700 + i) for i, c in enumerate(order)]
        rows = [np.zeros(100, ROW_DTYPE) for _ in frames]
        passes = [np.full(100, 9, np.uint8) for _ in frames]
        e.detect_batch(frames, rows, cams=[cams[c][0] for c in order], out_pass=passes)
        hits = 0
        for i, c in enumerate(order):
            plain = rows[i].copy()
            plain["zones"] = 0
            want_pass, want_zones = oracle_verdict(cams[c][3], plain)
            np.testing.assert_array_equal(passes[i], want_pass)
            np.testing.assert_array_equal(rows[i]["zones"], want_zones)
            assert (rows[i]["label"] >= 1).all() and (rows[i]["x_max"] < cams[c][1]).all() and (rows[i]["y_max"] < cams[c][2]).all()
            hits += int(want_pass.sum())
        assert hits > 0
        for f in flts:
            f.close()
    finally:
        e.close()


def detections_of(rows):
    out = []
    for r in rows:
        d = Detection(label=int(r["label"]), confidence=float(r["confidence"]),
                      bounding_box=BoundingBox(int(r["x_min"]), int(r["y_min"]), int(r["x_max"]), int(r["y_max"])))
        out.append(d)
    return out


def test_drop_mode_feeds_the_native_sieve(eng):
    """SURVEY 8(f)-1 end to end: a camera in drop mode + `HipTrackFilter().sieve()` on the rows as they leave the GPU
    == the reference's sieve (`TrackFilter([Confidence, Area, Mask])`, restated by the oracle) on the raw rows."""
    from oracle.tracker import TrackFilter, ZERO_ROW, sieve_rows
    from watsor_amd.filter.track import HipTrackFilter
    cfg = dict(golden_config()["config"])
    # every class the seeded random-init network reports may pass on confidence; area and the porch zones decide
    cfg["detect"] = [{name: {"area": 2, "confidence": 2, "zones": []}} for name in dict.fromkeys(COCO_CLASSES[1:])]
    alpha = porch_alpha()
    raw_cam = HipCameraFilter(eng, 6, cfg, alpha=alpha)
    drop_cam = HipCameraFilter(eng, 7, cfg, alpha=alpha, drop=True)
    oracle = TrackFilter([of.ConfidenceFilter(cfg), of.AreaFilter(cfg), of.MaskFilter(cfg, alpha=alpha)], 2, 4)
    native = HipTrackFilter(sensitivity=2, history=4)
    reported = 0
    for i in range(8):
        frame = synthetic_frame(640, 480, 40 + (i // 3))                 # a scene that changes every third frame
        raw, dropped = np.zeros(100, ROW_DTYPE), np.zeros(100, ROW_DTYPE)
        p_raw, p_drop = np.zeros(100, np.uint8), np.zeros(100, np.uint8)
        eng.detect_batch([frame, frame], [raw, dropped], cams=[6, 7], out_pass=[p_raw, p_drop])
        np.testing.assert_array_equal(p_raw, p_drop)
        keep = p_raw.astype(bool)
        assert keep.any() and not keep.all()
        assert dropped[keep].tobytes() == raw[keep].tobytes()
        assert dropped[~keep].tobytes() == bytes(72 * int((~keep).sum()))   # failing rows: all-zero records
        plain = raw.copy()
        plain["zones"] = 0                                                # the rows as the detector wrote them
        want, want_s = sieve_rows([oracle], detections_of(plain))
        got_s = native.sieve(dropped)
        assert got_s == want_s
        for k in range(100):
            w = want[k]
            g = dropped[k]
            assert (int(g["label"]), tuple(int(z) for z in g["zones"]), float(g["confidence"]),
                    (int(g["x_min"]), int(g["y_min"]), int(g["x_max"]), int(g["y_max"]))) == tuple(w), (i, k)
        reported += sum(1 for w in want if w != ZERO_ROW)
    assert reported > 0
    eng.set_camera_drop(7, False)
    raw, dropped = np.zeros(100, ROW_DTYPE), np.zeros(100, ROW_DTYPE)
    eng.detect_batch([frame, frame], [raw, dropped], cams=[6, 7])
    assert raw.tobytes() == dropped.tobytes()
    with pytest.raises(ValueError):
        eng.set_camera_drop(9, True)                                      # no filter on that camera
    raw_cam.close()
    drop_cam.close()


def test_mask_without_zones_rejects_everything(eng):
    """A configured mask whose alpha plane has no fully opaque pixel holds no polygon: the reference's MaskFilter then
    returns False for every detection (mask.py:44-59) -- it must not be taken for "no mask"."""
    alpha = np.full((100, 100), 200, np.uint8)
    cfg = {"width": 100, "height": 100, "detect": [{"person": {"area": 0, "confidence": 0, "zones": []}}]}
    assert len(oz.zone_polygons(alpha)) == 0
    flt = HipCameraFilter(eng, 2, cfg, alpha=alpha)
    assert flt.num_zones == 0
    rows = rows_from([(1, 0.70, (20, 20, 40, 80)), (1, 0.99, (0, 0, 99, 99))])
    want, _ = oracle_verdict([of.ConfidenceFilter(cfg), of.AreaFilter(cfg), of.MaskFilter(cfg, alpha=alpha)], rows.copy())
    got = flt.filter_rows(rows)
    assert not want.any() and not got.any() and not rows["zones"].any()
    flt.close()
    nomask = HipCameraFilter(eng, 2, cfg)                      # no mask at all: the same rows pass
    assert nomask.filter_rows(rows_from([(1, 0.70, (20, 20, 40, 80))]))[0] == 1
    nomask.close()
