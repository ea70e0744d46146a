"""The row comparison itself (oracle/compare.py): tolerances, and that a row without a partner passes only with a reason that
follows from the score tolerance.  Synthetic rows, no GPU."""
import numpy as np
import pytest

from oracle import compare as cmp
from watsor_amd.runtime import ROW_DTYPE


def _rows(items, n=100):
    r = np.zeros(n, ROW_DTYPE)
    for i, (label, score, box) in enumerate(items):
        r["label"][i], r["confidence"][i] = label, score
        r["x_min"][i], r["y_min"][i], r["x_max"][i], r["y_max"][i] = box
    return r


def _ref(items, n=100):
    lab = np.zeros(n, np.int32); conf = np.zeros(n, np.float64); box = np.zeros((n, 4), np.int32)
    for i, (label, score, b) in enumerate(items):
        lab[i], conf[i], box[i] = label, score, b
    return {"label": lab, "confidence": conf, "box": box}


BASE = [(1, 0.9, (10, 10, 110, 210)), (3, 0.8, (300, 40, 420, 200)), (1, 0.5, (200, 200, 320, 330))]


def test_stated_box_tolerance():
    assert cmp.box_tolerance_px(640, 480) == 1 and cmp.box_tolerance_px(1280, 720) == 2 and cmp.box_tolerance_px(1920, 1080) == 2
    assert cmp.box_tolerance_px(100, 100) == 1 and cmp.box_tolerance_px(3840, 2160) == 4


def test_identical_rows_match_with_zero_deltas():
    r = cmp.assert_rows_match(_rows(BASE), _ref(BASE), (480, 640, 3))
    assert len(r["pairs"]) == 3 and r["max_dscore"] == 0 and r["max_dbox_px"] == 0 and not r["missing"] and not r["extra"]


def test_score_and_box_deltas_are_measured_and_bounded():
    moved = [(1, 0.9004, (11, 10, 110, 212)), BASE[1], BASE[2]]
    r = cmp.compare_rows(_rows(moved), _ref(BASE), (1080, 1920, 3))
    assert r["max_dbox_px"] == 2 and abs(r["max_dscore"] - 4e-4) < 1e-9
    cmp.assert_rows_match(_rows(moved), _ref(BASE), (1080, 1920, 3))          # 2 px: inside the tolerance of a 1920x1080 frame ...
    with pytest.raises(AssertionError, match="dbox"):
        cmp.assert_rows_match(_rows(moved), _ref(BASE), (480, 640, 3))        # ... outside that of a 640x480 one
    with pytest.raises(AssertionError, match="dbox"):
        cmp.assert_rows_match(_rows([(1, 0.9, (13, 10, 110, 210)), BASE[1], BASE[2]]), _ref(BASE), (1080, 1920, 3))
    with pytest.raises(AssertionError, match="dscore"):
        cmp.assert_rows_match(_rows([(1, 0.902, (10, 10, 110, 210)), BASE[1], BASE[2]]), _ref(BASE), (480, 640, 3))


def test_a_row_that_is_simply_absent_is_not_tolerated():
    with pytest.raises(AssertionError, match="without an explanation"):
        cmp.assert_rows_match(_rows(BASE[:2]), _ref(BASE), (480, 640, 3))
    with pytest.raises(AssertionError, match="without an explanation"):     # ... and neither is one the oracle does not have
        cmp.assert_rows_match(_rows(BASE + [(7, 0.4, (5, 5, 50, 50))]), _ref(BASE), (480, 640, 3))


def test_nms_order_flip_is_explained():
    """Two same-class boxes with IoU > 0.6 and scores 3e-4 apart: the oracle keeps A and suppresses B, a detector whose scores moved by
    the tolerance keeps B and suppresses A.  Each side's odd row is explained by the other's."""
    a, b = (1, 0.5003, (200, 200, 320, 330)), (1, 0.5001, (204, 206, 326, 334))
    assert cmp.box_iou_px(a[2], b[2]) > 0.6
    r = cmp.assert_rows_match(_rows(BASE[:2] + [b]), _ref(BASE[:2] + [a]), (480, 640, 3))
    assert len(r["missing"]) == 1 and "NMS order" in r["missing"][0][1] and len(r["extra"]) == 1 and "NMS order" in r["extra"][0][1]
    far = (1, 0.48, b[2])                                   # the same neighbour 2e-2 below: no longer a tie within the tolerance
    with pytest.raises(AssertionError, match="without an explanation"):
        cmp.assert_rows_match(_rows(BASE[:2] + [far]), _ref(BASE[:2] + [a]), (480, 640, 3))


def test_top_100_cut_is_explained_only_at_the_cut():
    items = [(1 + i % 5, 0.9 - 0.005 * i, (5 * i, 3 * i, 5 * i + 60, 3 * i + 90)) for i in range(100)]
    last = items[99]
    other = (9, last[1] + 2e-4, (400, 300, 500, 420))       # what was 101st for the oracle is 100th for the detector
    r = cmp.assert_rows_match(_rows(items[:99] + [other]), _ref(items), (480, 640, 3))
    assert "cut" in r["missing"][0][1] and "cut" in r["extra"][0][1]
    high = (9, 0.7, (400, 300, 500, 420))                   # a row in the middle of the list is not "at the cut"
    got = sorted(items[:99] + [high], key=lambda t: -t[1])
    with pytest.raises(AssertionError, match="without an explanation"):
        cmp.assert_rows_match(_rows(got), _ref(items), (480, 640, 3))
