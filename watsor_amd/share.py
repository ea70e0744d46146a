"""ABI of the result buffer: ctypes mirrors of the reference's shared-memory structs.

Byte-for-byte the layout of `watsor/stream/share.py:11-32` (`BoundingBox` 16 B, `Detection` 72 B
with `label@0, zones[10]@4, confidence(double)@48, bounding_box@56`, `Header` 7224 B with
`detections@24`).  The HIP library writes `Detection[100]` rows in place through these offsets
(`include/watsor_hip.h: wz_detection_t`), so a reference `Frame.header.detections` array can be
handed to `HipObjectDetector.detect()` unchanged; these classes exist so that the package also
works (tests, bench, smoke) where the reference package is not installed.
"""
from ctypes import Structure, c_double, c_int, sizeof

MAX_DETECTIONS = 100
MAX_ZONES = 10


class BoundingBox(Structure):
    _fields_ = [('x_min', c_int),
                ('y_min', c_int),
                ('x_max', c_int),
                ('y_max', c_int)]


class Detection(Structure):
    _fields_ = [('label', c_int),
                ('zones', c_int * MAX_ZONES),
                ('confidence', c_double),
                ('bounding_box', BoundingBox)]


class Header(Structure):
    _fields_ = [('width', c_int),
                ('height', c_int),
                ('channels', c_int),
                ('epoch', c_double),
                ('detections', Detection * MAX_DETECTIONS)]


assert sizeof(BoundingBox) == 16 and sizeof(Detection) == 72 and sizeof(Header) == 7224
assert Detection.confidence.offset == 48 and Detection.bounding_box.offset == 56 and Header.detections.offset == 24

DetectionArray = Detection * MAX_DETECTIONS
