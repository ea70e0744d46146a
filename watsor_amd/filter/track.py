"""`TrackFilter` of the reference (`watsor/filter/track.py:8-149`) on the native tracker.

Same constructor (`filters=None, sensitivity=5, history=10`) and the same call protocol
(`detections -> (list of Detection, suspicious_activity)`), so it drops into `DetectionSieve`'s filter list
(`watsor/main.py:294-299`, `watsor/filter/sieve.py:24-27`).  The grouping / matching / combining runs in
`wz_tracker_update` (csrc/wz_tracker.cpp); rows are handed over as one contiguous `Detection` array.

Two ways to use it:

  * `HipTrackFilter(filters)` -- the per-detection filters are the reference's Python callables and are
    evaluated here exactly like `track.py:26` (`label > 0 and all(f(d) for f in filters)`, short-circuit, on the
    sieve's clones so that MaskFilter's zone side effect is kept).
  * `HipTrackFilter()` behind a camera whose filters run on the GPU (`HipCameraFilter(..., drop=True)`): failing rows
    arrive as all-zero rows, `label > 0` is the only test left, and `sieve()` processes the frame header's
    `Detection[100]` in place with a single native call and no per-row Python at all
    (`DetectionSieve._incoming_frame`, sieve.py:21-33).

There is no Python fallback: the tracker lives in libwatsor_hip.so (`watsor_amd/_lib.py` raises if it is missing).
"""
from __future__ import annotations

import ctypes as C

from .. import _lib
from ..share import MAX_DETECTIONS, Detection


class HipTrackFilter(object):
    def __init__(self, filters=None, sensitivity=5, history=10):
        self.__filters = [] if filters is None else filters
        self.__sensitivity, self.__history = sensitivity, history
        self.__lib = _lib.load()
        handle = C.c_void_p()
        _lib.check(self.__lib.wz_tracker_create(int(sensitivity), int(history), C.byref(handle)), "wz_tracker_create")
        self.__handle = handle
        self.__in = (Detection * MAX_DETECTIONS)()
        self.__out = (Detection * MAX_DETECTIONS)()

    def __del__(self):
        handle, self.__handle = getattr(self, "_HipTrackFilter__handle", None), None
        if handle:
            self.__lib.wz_tracker_destroy(handle)

    @property
    def tracks(self) -> int:
        return self.__lib.wz_tracker_count(self.__handle)

    def reset(self):
        _lib.check(self.__lib.wz_tracker_reset(self.__handle))

    def __call__(self, detections):
        """track.py:25-27 + 29-110.  `detections`: iterable of Detection structs (the reference's or ours)."""
        n = 0
        size = C.sizeof(Detection)
        for d in detections:
            if d.label > 0 and all(f(d) for f in self.__filters):
                if n == len(self.__in):                        # more than 100 rows: not the sieve, grow
                    bigger = (Detection * (2 * n))()
                    C.memmove(bigger, self.__in, n * size)
                    self.__in = bigger
                C.memmove(C.byref(self.__in, n * size), C.addressof(d), size)
                n += 1
        if len(self.__out) < n:
            self.__out = (Detection * len(self.__in))()
        n_out, suspicious = C.c_int(0), C.c_int(0)
        _lib.check(self.__lib.wz_tracker_update(self.__handle, self.__in, n, None, self.__out, len(self.__out),
                                                C.byref(n_out), C.byref(suspicious)), "wz_tracker_update")
        result = []
        for i in range(min(n_out.value, len(self.__out))):
            row = Detection()
            C.memmove(C.addressof(row), C.byref(self.__out, i * size), size)
            result.append(row)
        return result, bool(suspicious.value)

    def sieve(self, detections, passed=None) -> bool:
        """`DetectionSieve._incoming_frame` (sieve.py:21-33) for a sieve whose only filter is this tracker, in
        place on a ctypes `Detection` array (e.g. `frame.header.detections`) or a numpy `ROW_DTYPE` array; `passed`: optional 100 pass bytes
        (numpy uint8) from `HipEngine.collect` / `detect_batch`.  Returns suspicious_activity."""
        if self.__filters:
            raise ValueError("sieve() is the pre-filtered fast path: construct HipTrackFilter() without filters")
        n = len(detections)
        if hasattr(detections, "ctypes"):                          # numpy rows (runtime.ROW_DTYPE)
            if detections.dtype.itemsize != C.sizeof(Detection) or not detections.flags.c_contiguous:
                raise ValueError("rows must be contiguous 72-byte Detection records")
            address = detections.ctypes.data
        else:
            address = C.addressof(detections)
        p = None
        if passed is not None:
            if passed.dtype.itemsize != 1 or passed.size < n or not passed.flags.c_contiguous:
                raise ValueError("passed must be %d contiguous bytes" % n)
            p = C.c_void_p(passed.ctypes.data)
        suspicious = C.c_int(0)
        _lib.check(self.__lib.wz_tracker_sieve(self.__handle, C.c_void_p(address), n, p,
                                               C.byref(suspicious)), "wz_tracker_sieve")
        return bool(suspicious.value)
