"""Row ABI: our ctypes mirror == the reference's shared-memory structs (watsor/stream/share.py:11-32)."""
import ctypes
import re
import os

import numpy as np
import pytest

from watsor_amd import share
from watsor_amd.runtime import ROW_DTYPE


def test_sizes_and_offsets():
    assert ctypes.sizeof(share.BoundingBox) == 16
    assert ctypes.sizeof(share.Detection) == 72
    assert ctypes.sizeof(share.Header) == 7224
    assert share.Detection.label.offset == 0
    assert share.Detection.zones.offset == 4
    assert share.Detection.confidence.offset == 48
    assert share.Detection.bounding_box.offset == 56
    assert share.Header.detections.offset == 24


def test_numpy_row_dtype_matches_ctypes():
    assert ROW_DTYPE.itemsize == 72
    assert ROW_DTYPE.fields["confidence"][1] == 48
    assert ROW_DTYPE.fields["x_min"][1] == 56 and ROW_DTYPE.fields["y_max"][1] == 68
    rows = share.DetectionArray()
    rows[3].label = 7
    rows[3].confidence = 0.25
    rows[3].bounding_box.x_max = 99
    rows[3].zones[9] = 5
    a = np.frombuffer(rows, dtype=ROW_DTYPE)
    assert a[3]["label"] == 7 and a[3]["confidence"] == 0.25 and a[3]["x_max"] == 99 and a[3]["zones"][9] == 5


def test_c_header_struct_matches():
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "watsor_hip.h")).read()
    assert "WZ_MAX_DETECTIONS 100" in hdr and "WZ_MAX_ZONES 10" in hdr and "WZ_NUM_LABELS 91" in hdr
    body = re.search(r"typedef struct wz_detection \{(.*?)\} wz_detection_t;", hdr, re.S).group(1)
    fields = [l.strip() for l in body.strip().splitlines()]
    assert fields == ["int32_t label;", "int32_t zones[WZ_MAX_ZONES];", "int32_t _pad;", "double confidence;",
                      "int32_t x_min, y_min, x_max, y_max;"]


@pytest.mark.reference
def test_against_reference_structs(reference_on_path):
    from watsor.stream import share as ref
    for name in ("BoundingBox", "Detection", "Header"):
        ours, theirs = getattr(share, name), getattr(ref, name)
        assert ctypes.sizeof(ours) == ctypes.sizeof(theirs)
        for (fn, _), (rn, _) in zip(ours._fields_, theirs._fields_):
            assert fn == rn
            assert getattr(ours, fn).offset == getattr(theirs, rn).offset
            assert getattr(ours, fn).size == getattr(theirs, rn).size
    # a reference Detection array is accepted wherever ours is (same memory layout)
    arr = (ref.Detection * 100)()
    a = np.frombuffer(arr, dtype=ROW_DTYPE)
    arr[1].bounding_box.y_min = 11
    assert a[1]["y_min"] == 11
