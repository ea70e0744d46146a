"""Generates tests/golden/track.json by running the REFERENCE's own `TrackFilter`
(`watsor/filter/track.py`, imported from /root/reference) over seeded multi-frame sequences.

Run in the build container only:   python tests/golden/make_track_golden.py
The fixture travels to the GPU box; /root/reference does not.

`np.argsort` in `track.py:67` is called with the default (unstable) kind; the numpy in this image sorts with
AVX-512 networks and permutes EQUAL keys, so which of two equally-near tracks claims a contested detection is
defined by the numpy build, not by the reference.  Every sequence is therefore run twice: with the reference
untouched, and with the reference's module-level `np` wrapped so that this one call sorts with kind="stable".
Where both runs agree on every frame the sequence is tagged `"pinned": "reference"`; where they differ the
stable run is stored and tagged `"pinned": "reference, argsort kind=stable"` (+ the number of differing frames).

Each sequence = constructor arguments + per frame the input rows and the rows / suspicious flag the reference
returned.  Rows are [label, [zones x10], confidence, [x_min, y_min, x_max, y_max]].
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

import watsor.filter.track as reference_track                # noqa: E402  (reference)
from watsor.filter.track import TrackFilter                 # noqa: E402  (reference)
from watsor.stream.share import BoundingBox, Detection      # noqa: E402  (reference)


def to_detection(r):
    d = Detection(label=r[0], confidence=r[2], bounding_box=BoundingBox(*r[3]))
    for i, z in enumerate(r[1]):
        d.zones[i] = z
    return d


def to_row(d):
    bb = d.bounding_box
    return [int(d.label), [int(z) for z in d.zones], float(d.confidence),
            [int(bb.x_min), int(bb.y_min), int(bb.x_max), int(bb.y_max)]]


def zones_of(rng, n_zones):
    zs = [0] * 10
    if n_zones:
        k = int(rng.integers(0, 4))
        picks = rng.choice(np.arange(1, n_zones + 1), size=min(k, n_zones), replace=False)
        for i, z in enumerate(picks):
            zs[i] = int(z)
    return zs


def scene(rng, frames, labels, max_objects, width, height, jitter, p_miss, p_new, n_zones, grid=1, with_zero=True):
    """Objects drifting with jitter; some frames miss an object, new ones appear; label-0 padding rows mixed in."""
    objects = []
    out = []
    for _ in range(frames):
        if len(objects) < max_objects and (not objects or rng.random() < p_new):
            for _ in range(int(rng.integers(1, 4))):
                w, h = int(rng.integers(8, width // 3)), int(rng.integers(8, height // 3))
                objects.append(dict(label=int(rng.choice(labels)), x=int(rng.integers(0, width - w)),
                                    y=int(rng.integers(0, height - h)), w=w, h=h,
                                    vx=int(rng.integers(-3, 4)), vy=int(rng.integers(-3, 4))))
        rows = []
        for o in objects:
            o["x"] += o["vx"]
            o["y"] += o["vy"]
            if rng.random() < p_miss:
                continue
            jx, jy, jw, jh = (int(v) for v in rng.integers(-jitter, jitter + 1, 4))
            x0, y0 = (o["x"] + jx) // grid * grid, (o["y"] + jy) // grid * grid
            x1, y1 = x0 + max(1, o["w"] + jw) // grid * grid, y0 + max(1, o["h"] + jh) // grid * grid
            rows.append([o["label"], zones_of(rng, n_zones), float(np.float32(rng.uniform(0.3, 1.0))), [x0, y0, x1, y1]])
        if rng.random() < 0.1 and objects:
            objects.pop(int(rng.integers(0, len(objects))))
        order = rng.permutation(len(rows))
        rows = [rows[i] for i in order]
        if with_zero:
            rows += [[0, [0] * 10, 0.0, [0, 0, 0, 0]]] * int(rng.integers(0, 3))
        out.append(rows)
    return out


def crowd(rng, frames, n, labels, width, height, grid):
    """Many same-label boxes per frame on a coarse grid: equal distances, contested nearest inputs, large sets."""
    out = []
    for _ in range(frames):
        rows = []
        for _ in range(int(rng.integers(max(1, n // 2), n + 1))):
            x0, y0 = int(rng.integers(0, width // grid)) * grid, int(rng.integers(0, height // grid)) * grid
            rows.append([int(rng.choice(labels)), zones_of(rng, 12), float(np.float32(rng.uniform(0.3, 1.0))),
                         [x0, y0, x0 + grid * int(rng.integers(1, 4)), y0 + grid * int(rng.integers(1, 4))]])
        out.append(rows)
    return out


class StableArgsortNumpy(object):
    """numpy with `argsort` defaulting to the stable kind; everything else passes through."""

    def __getattr__(self, item):
        return getattr(np, item)

    @staticmethod
    def argsort(a, *args, **kwargs):
        return np.argsort(a, kind="stable")


def run_once(frames, sensitivity, history):
    flt = TrackFilter(sensitivity=sensitivity, history=history)
    rec = []
    for rows in frames:
        result, suspicious = flt([to_detection(r) for r in rows])
        rec.append(dict(rows=rows, out=[to_row(d) for d in result], suspicious=bool(suspicious)))
    return rec


def run(name, frames, sensitivity, history):
    plain = run_once(frames, sensitivity, history)
    reference_track.np = StableArgsortNumpy()
    try:
        stable = run_once(frames, sensitivity, history)
    finally:
        reference_track.np = np
    differing = sum(1 for a, b in zip(plain, stable) if a != b)
    pinned = "reference" if differing == 0 else "reference, argsort kind=stable"
    print("%-24s %3d frames  %s%s" % (name, len(frames), pinned, "" if not differing else " (%d frames differ)" % differing))
    return dict(name=name, sensitivity=sensitivity, history=history, pinned=pinned, differing_frames=differing,
                frames=stable)


def main():
    rng = np.random.Generator(np.random.PCG64(77))
    seqs = []
    # the reference's known-answer test, watsor/test/test_filter.py:76-97
    kat = [[[1, [0] * 10, 0.70, [50, 50, 60, 60]], [1, [0] * 10, 0.70, [10, 10, 30, 30]]],
           [[1, [0] * 10, 0.70, [40, 40, 55, 55]], [1, [0] * 10, 0.70, [80, 80, 90, 90]]]]
    seqs.append(run("kat_test_filter", kat, 1, 2))
    seqs.append(run("defaults_street", scene(rng, 60, [1, 3, 8], 6, 640, 480, 3, 0.1, 0.15, 0), 5, 10))
    seqs.append(run("zones_porch", scene(rng, 60, [1, 2, 17], 8, 1280, 720, 5, 0.05, 0.2, 5), 3, 10))
    seqs.append(run("flapping", scene(rng, 40, [1], 4, 320, 240, 12, 0.0, 0.3, 3), 2, 4))
    seqs.append(run("history_1", scene(rng, 30, [1, 3], 5, 640, 480, 2, 0.2, 0.3, 0), 1, 1))
    seqs.append(run("never_reported", scene(rng, 20, [1], 3, 640, 480, 2, 0.0, 0.3, 0), 6, 3))
    seqs.append(run("negative_coordinates", [[[1, [0] * 10, 0.5, [-7, -3, 2, 4]]], [[1, [0] * 10, 0.6, [-9, -5, 0, 2]]],
                                             [[1, [0] * 10, 0.4, [-3, -3, -2, 5]]]], 1, 3))
    seqs.append(run("empty_frames", [[], kat[0], [], [], kat[1], [[0, [0] * 10, 0.0, [0, 0, 0, 0]]] * 3, kat[0]], 1, 2))
    seqs.append(run("crowd_grid_12", crowd(rng, 25, 12, [1], 640, 480, 40), 2, 5))
    seqs.append(run("crowd_grid_40", crowd(rng, 25, 40, [1, 3], 640, 480, 20), 2, 5))
    seqs.append(run("crowd_100", crowd(rng, 12, 100, [1], 1920, 1080, 8), 1, 3))
    seqs.append(run("crowd_free_30", crowd(rng, 30, 30, [1, 2], 1280, 720, 1), 2, 4))
    seqs.append(run("crowd_free_100", crowd(rng, 15, 100, [1], 1920, 1080, 1), 1, 3))
    seqs.append(run("crowd_sparse_labels", crowd(rng, 30, 30, list(range(1, 20)), 640, 480, 16), 1, 3))
    json.dump(seqs, open(os.path.join(HERE, "track.json"), "w"), separators=(",", ":"))
    print("wrote track.json: %d sequences, %d frames, %d KiB" % (
        len(seqs), sum(len(s["frames"]) for s in seqs), os.path.getsize(os.path.join(HERE, "track.json")) // 1024))


if __name__ == "__main__":
    main()
