"""the sieve's tracker.  Oracle (oracle/tracker.py) and native tracker (csrc/wz_tracker.cpp through
the C ABI) against tests/golden/track.json -- sequences recorded from the REFERENCE's own `TrackFilter`
(tests/golden/make_track_golden.py) -- plus the known-answer test of `watsor/test/test_filter.py:76-97`,
the CPython-set order emulation against real sets, and seeded fuzzing of native vs oracle.  No GPU involved.
"""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

from oracle import filters as of
from oracle.tracker import Row, TrackFilter, ZERO_ROW, sieve_rows
from watsor_amd import _lib
from watsor_amd.filter.track import HipTrackFilter
from watsor_amd.share import BoundingBox, Detection, DetectionArray

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "track.json")


@pytest.fixture(scope="module")
def sequences():
    return json.load(open(GOLDEN))


def detection_of(r):
    d = Detection(label=r[0], confidence=r[2], bounding_box=BoundingBox(*r[3]))
    for i, z in enumerate(r[1]):
        d.zones[i] = z
    return d


def listed(d):
    if isinstance(d, Row):
        return [d.label, list(d.zones), d.confidence, list(d.box)]
    bb = d.bounding_box
    return [d.label, list(d.zones), d.confidence, [bb.x_min, bb.y_min, bb.x_max, bb.y_max]]


def test_golden_covers_the_cases(sequences):
    names = {s["name"] for s in sequences}
    assert {"kat_test_filter", "defaults_street", "zones_porch", "empty_frames", "crowd_100", "crowd_free_100"} <= names
    # most sequences are pinned by the untouched reference; the tie-heavy ones by the reference with a stable argsort
    assert sum(1 for s in sequences if s["pinned"] == "reference") >= 10
    assert any(s["pinned"] != "reference" for s in sequences)
    assert sum(len(f["out"]) for s in sequences for f in s["frames"]) > 3000


def test_oracle_equals_reference_recordings(sequences):
    for s in sequences:
        flt = TrackFilter(sensitivity=s["sensitivity"], history=s["history"])
        for i, fr in enumerate(s["frames"]):
            out, suspicious = flt([Row(r[0], tuple(r[1]), r[2], tuple(r[3])) for r in fr["rows"]])
            assert [listed(o) for o in out] == fr["out"], (s["name"], i)
            assert suspicious == fr["suspicious"], (s["name"], i)


def test_native_tracker_equals_reference_recordings(sequences):
    for s in sequences:
        flt = HipTrackFilter(sensitivity=s["sensitivity"], history=s["history"])
        for i, fr in enumerate(s["frames"]):
            out, suspicious = flt([detection_of(r) for r in fr["rows"]])
            assert [listed(o) for o in out] == fr["out"], (s["name"], i)
            assert suspicious == fr["suspicious"], (s["name"], i)


def test_native_sieve_in_place_equals_reference_recordings(sequences):
    """sieve.py:21-33,44-56: survivors first, the other rows zeroed (padding bytes included)."""
    for s in sequences:
        flt = HipTrackFilter(sensitivity=s["sensitivity"], history=s["history"])
        for i, fr in enumerate(s["frames"]):
            if len(fr["rows"]) > 100:
                continue
            rows = DetectionArray()
            for k, r in enumerate(fr["rows"]):
                rows[k] = detection_of(r)
            suspicious = flt.sieve(rows)
            expect = fr["out"][:100] + [listed(ZERO_ROW)] * (100 - min(100, len(fr["out"])))
            assert [listed(d) for d in rows] == expect, (s["name"], i)
            assert suspicious == fr["suspicious"]
            raw = bytes(rows)
            assert all(raw[72 * k + 44:72 * k + 48] == b"\0\0\0\0" for k in range(100))


def test_known_answer_of_the_reference_test():
    """watsor/test/test_filter.py:76-97, literally."""
    flt = HipTrackFilter(sensitivity=1, history=2)
    out, suspicious = flt([Detection(label=1, confidence=0.70, bounding_box=BoundingBox(50, 50, 60, 60)),
                           Detection(label=1, confidence=0.70, bounding_box=BoundingBox(10, 10, 30, 30))])
    assert suspicious and len(out) == 2
    assert listed(out[0])[3] == [50, 50, 60, 60] and listed(out[1])[3] == [10, 10, 30, 30]
    out, suspicious = flt([Detection(label=1, confidence=0.70, bounding_box=BoundingBox(40, 40, 55, 55)),
                           Detection(label=1, confidence=0.70, bounding_box=BoundingBox(80, 80, 90, 90))])
    assert suspicious and len(out) == 2
    assert listed(out[0])[3] == [40, 40, 60, 60] and listed(out[1])[3] == [80, 80, 90, 90]


def test_cpython_set_order_emulation():
    """New tracks and combined zones follow `set` iteration order (track.py:89,97,136-146).  (The two hooks into the
    set emulation are exported by the development library only.)"""
    lib = _lib.load(dev=True)
    rng = random.Random(5)
    out = np.zeros(128, np.int32)
    scrambled = 0
    for _ in range(3000):
        n = rng.randint(0, 60)
        hi = rng.choice([10, 40, 100, 1000, 70000])
        keys = [rng.randint(0, hi) for _ in range(n)]
        real = set()
        for k in keys:
            real.add(k)
        a = np.array(keys + [0], np.int32)
        m = lib.wz_debug_pyset_order(a.ctypes.data, n, out.ctypes.data)
        assert list(out[:m]) == list(real)
        scrambled += list(real) != sorted(real)
    for _ in range(3000):
        n = rng.randint(0, 100)
        p = rng.random()
        used = [rng.random() < p for _ in range(n)]
        real = set(range(n)).difference({i for i in range(n) if used[i]})
        a = np.array(used + [0], np.uint8)
        m = lib.wz_debug_unused_order(n, a.ctypes.data, out.ctypes.data)
        assert list(out[:m]) == list(real)
        scrambled += list(real) != sorted(real)
    assert scrambled > 100                     # the cases where the order is not simply ascending were exercised


def random_rows(rng, n, labels, span, n_zones):
    rows = []
    for _ in range(n):
        x0, y0 = int(rng.integers(-5, span)), int(rng.integers(-5, span))
        zones = [0] * 10
        for k in range(int(rng.integers(0, 4))):
            zones[k] = int(rng.integers(1, n_zones + 1))
        rows.append([int(rng.choice(labels)), zones, float(np.float32(rng.random())),
                     [x0, y0, x0 + int(rng.integers(0, 30)), y0 + int(rng.integers(0, 30))]])
    return rows


@pytest.mark.parametrize("seed,n_max,labels,span,sens,hist", [
    (1, 6, [1, 2, 3], 60, 1, 3), (2, 30, [1], 40, 2, 4), (3, 100, [0, 1, 5, 90], 25, 1, 10),
    (4, 100, [7], 12, 3, 5), (5, 12, list(range(0, 30)), 300, 1, 1)])
def test_native_equals_oracle_fuzz(seed, n_max, labels, span, sens, hist):
    """Coarse coordinates: many equal distances and contested detections; both sides break ties by index."""
    rng = np.random.Generator(np.random.PCG64(seed))
    oracle, native = TrackFilter(sensitivity=sens, history=hist), HipTrackFilter(sensitivity=sens, history=hist)
    for frame in range(60):
        rows = random_rows(rng, int(rng.integers(0, n_max + 1)), labels, span, 14)
        exp, exp_s = oracle([Row(r[0], tuple(r[1]), r[2], tuple(r[3])) for r in rows])
        got, got_s = native([detection_of(r) for r in rows])
        assert [listed(d) for d in got] == [listed(d) for d in exp], frame
        assert got_s == exp_s
        assert native.tracks == sum(len(v) for v in oracle.tracks.values())


CONFIG = {"width": 640, "height": 480,
          "detect": [{"person": {"area": 1, "confidence": 40, "zones": []}},
                     {"car": {"area": 2, "confidence": 60, "zones": []}}]}


def test_python_filters_in_front_like_the_reference():
    """track.py:26 with the reference-style callables (here the oracle's restatements of them)."""
    filters = [of.ConfidenceFilter(CONFIG), of.AreaFilter(CONFIG)]
    oracle, native = TrackFilter(filters, 2, 5), HipTrackFilter(filters, 2, 5)
    rng = np.random.Generator(np.random.PCG64(11))
    for frame in range(40):
        rows = random_rows(rng, int(rng.integers(0, 20)), [0, 1, 3, 8], 400, 3)
        for r in rows:
            r[3][2] += int(rng.integers(0, 200))
            r[3][3] += int(rng.integers(0, 200))
        exp, exp_s = oracle([detection_of(r) for r in rows])
        got, got_s = native([detection_of(r) for r in rows])
        assert [listed(d) for d in got] == [listed(d) for d in exp]
        assert got_s == exp_s


def test_pass_bytes_equal_filtering_first():
    """`wz_tracker_sieve(..., pass)` = the GPU's verdict bytes instead of the Python filters."""
    filters = [of.ConfidenceFilter(CONFIG), of.AreaFilter(CONFIG)]
    oracle, native = TrackFilter(filters, 1, 4), HipTrackFilter(sensitivity=1, history=4)
    rng = np.random.Generator(np.random.PCG64(12))
    for frame in range(30):
        rows = random_rows(rng, 100, [0, 1, 3, 8], 400, 3)
        for r in rows:
            r[3][2] += int(rng.integers(0, 200))
            r[3][3] += int(rng.integers(0, 200))
        arr = DetectionArray()
        passed = np.zeros(100, np.uint8)
        for k, r in enumerate(rows):
            arr[k] = detection_of(r)
            passed[k] = all(f(arr[k]) for f in filters)          # the test's stand-in for wz_k_rows' pass byte
        exp, exp_s = sieve_rows([oracle], [detection_of(r) for r in rows])
        got_s = native.sieve(arr, passed)
        assert [listed(d) for d in arr] == [listed(d) for d in exp]
        assert got_s == exp_s


def test_errors_and_lifecycle():
    with pytest.raises(ValueError, match="history"):
        HipTrackFilter(history=0)
    flt = HipTrackFilter([lambda d: True])
    with pytest.raises(ValueError, match="pre-filtered"):
        flt.sieve(DetectionArray())
    flt = HipTrackFilter(sensitivity=1, history=2)
    out, suspicious = flt([])
    assert out == [] and suspicious is False and flt.tracks == 0
    flt([Detection(label=2, confidence=0.5, bounding_box=BoundingBox(1, 1, 5, 5))])
    assert flt.tracks == 1
    flt.reset()
    assert flt.tracks == 0
    with pytest.raises(ValueError):
        flt.sieve(DetectionArray(), np.zeros(10, np.uint8))
    lib = _lib.load()
    assert lib.wz_tracker_update(None, None, 0, None, None, 0, None, None) < 0
    assert b"wz_tracker_update" in lib.wz_last_error()
