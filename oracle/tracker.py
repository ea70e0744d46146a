"""Oracle (test infrastructure): the recurring-detection tracker and the sieve's row copy.

Restates `watsor/filter/track.py:25-149` (`TrackFilter`) and the copy-in / copy-out of
`watsor/filter/sieve.py:21-33,44-56` (`DetectionSieve._incoming_frame`, `_copy_from`, `_copy_to`) on plain
value rows, so that `csrc/wz_tracker.cpp` can be compared with it row for row.  Only `tests/` import this.

Pinned by the reference's known-answer test `watsor/test/test_filter.py:76-97` and by
`tests/golden/track.json`, which `tests/golden/make_track_golden.py` generates by running the REFERENCE's own
`TrackFilter` class (imported from /root/reference in the build container) on seeded sequences.

Two things in the reference are defined by the Python runtime rather than by the source, and are followed here
the same way:
  * the order in which unmatched detections become new tracks, and the order of a combined row's zones, is the
    iteration order of a CPython `set` of small ints (`track.py:92,97-99,136-146`).  This oracle uses real
    `set`s in the same sequence of operations, so under CPython it yields the reference's order;
  * `np.argsort` (`track.py:67`) is called with the default, unstable kind: rows whose nearest-input distances
    are EQUAL are visited in an order numpy does not define (it differs between numpy builds / CPU features).
    This oracle visits them in index order (`kind="stable"`).  Status: **tie order unpinned**; the golden
    sequences assert equality with the reference on every frame, ties included, for the numpy in this image.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Callable, Iterable, List, Sequence, Tuple

import numpy as np

MAX_ZONES = 10
MAX_DETECTIONS = 100

Row = namedtuple("Row", "label zones confidence box")          # zones: 10-tuple, box: (x_min, y_min, x_max, y_max)
ZERO_ROW = Row(0, (0,) * MAX_ZONES, 0.0, (0, 0, 0, 0))


def row_of(detection) -> Row:
    """Value copy of a `Detection` struct (sieve.py:38-42 `_clone`)."""
    if isinstance(detection, Row):
        return detection
    bb = detection.bounding_box
    return Row(int(detection.label), tuple(int(z) for z in detection.zones), float(detection.confidence),
               (int(bb.x_min), int(bb.y_min), int(bb.x_max), int(bb.y_max)))


def centre(box) -> Tuple[int, int]:
    """track.py:112-116: float division, truncation toward zero."""
    return int((box[0] + box[2]) / 2.0), int((box[1] + box[3]) / 2.0)


def merge(history: Sequence[Row]) -> Row:
    """track.py:118-149: label/confidence/box of the OLDEST row widened by the younger ones; zones = union."""
    first = history[0]
    conf = first.confidence
    x0, y0, x1, y1 = first.box
    for r in history[1:]:
        conf = max(conf, r.confidence)
        x0, y0 = min(x0, r.box[0]), min(y0, r.box[1])
        x1, y1 = max(x1, r.box[2]), max(y1, r.box[3])
    seen = set()
    for r in history:                                            # track.py:136-139
        for z in r.zones:
            if z > 0:
                seen.add(z)
    zones = list(seen)[:MAX_ZONES]                               # track.py:141-146 (set iteration order)
    zones += [0] * (MAX_ZONES - len(zones))
    return Row(first.label, tuple(zones), conf, (x0, y0, x1, y1))


class TrackFilter(object):
    def __init__(self, filters: Iterable[Callable] = None, sensitivity: int = 5, history: int = 10):
        self.sensitivity, self.history = sensitivity, history
        self.filters = [] if filters is None else list(filters)
        self.tracks = {}                                         # label -> list of histories (oldest row first)

    def __call__(self, detections):
        """detections: `Row`s, or mutable Detection-like objects (the sieve's clones) which the per-detection
        filters may write zones into (mask.py:52-57) before the value copy is taken."""
        passed = [row_of(d) for d in detections if d.label > 0 and all(f(d) for f in self.filters)]  # track.py:25-27
        return self.update(passed)

    def update(self, passed: Sequence[Row]):
        by_label = {}
        for r in passed:                                         # track.py:31-33
            by_label.setdefault(r.label, []).append(r)
        suspicious = len(by_label) > 0                           # track.py:38

        for label in [l for l in self.tracks if l not in by_label]:      # track.py:41-46
            del self.tracks[label]

        for label, inputs in by_label.items():
            known = self.tracks.setdefault(label, [])            # defaultdict access, track.py:56
            n_in, n_known = len(inputs), len(known)
            cin = np.array([centre(r.box) for r in inputs], dtype=np.int64).reshape(n_in, 2)
            ckn = np.array([centre(h[0].box) for h in known], dtype=np.int64).reshape(n_known, 2)
            order: List[int] = []
            nearest: List[int] = []
            if n_known > 0 and n_in > 0:
                # scipy's euclidean cdist (track.py:63): sqrt of the sum of squared differences in double
                d = ckn[:, None, :].astype(np.float64) - cin[None, :, :].astype(np.float64)
                dist = np.sqrt(d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1])
                order = [int(i) for i in np.argsort(dist.min(axis=1), kind="stable")]      # track.py:67
                nearest = [int(dist[i].argmin()) for i in order]                          # track.py:70
            used_k, used_i = set(), set()
            for k, i in zip(order, nearest):                     # track.py:77-86: each track claims only its nearest
                if k in used_k or i in used_i:
                    continue
                known[k].append(inputs[i])
                if len(known[k]) > self.history:                 # deque(maxlen=history), track.py:98
                    del known[k][0]
                used_k.add(k)
                used_i.add(i)
            lost = set(range(n_known)).difference(used_k)        # track.py:88-89
            fresh = set(range(n_in)).difference(used_i)
            for k in sorted(lost, reverse=True):                 # track.py:92-94
                del known[k]
            for i in fresh:                                      # track.py:97-99 (set iteration order)
                known.append([inputs[i]][-self.history:] if self.history > 0 else [])

        out = []
        for label, known in self.tracks.items():                 # track.py:103-110
            for h in known:
                if len(h) < self.sensitivity:
                    continue
                out.append(merge(h))
        return out, suspicious


def sieve_rows(track_filters: Sequence[TrackFilter], rows: Sequence[Row]) -> Tuple[List[Row], bool]:
    """sieve.py:21-33 on one frame's 100 rows: filters chained, result written back, the rest zeroed."""
    cur = list(rows)
    suspicious = False
    for flt in track_filters:
        cur, sa = flt(cur)
        suspicious |= sa
    out = list(cur[:len(rows)])
    out += [ZERO_ROW] * (len(rows) - len(out))
    return out, suspicious
