"""Per-frame cost of the sieve step: the reference's Python path vs the native one.

    python tools/sieve_bench.py [--frames 2000] [--json out.json]

Frames: the "defaults_street" / "zones_porch" golden scenes (tests/golden/track.json) looped, each padded to the 100
rows a detector writes (`tensorflow_cpu.py:79-90`: rows past the detections carry label 1, confidence 0).

  reference : `DetectionSieve._copy_from` -> `TrackFilter([ConfidenceFilter, AreaFilter])` -> `_copy_to`
              (watsor/filter/sieve.py:21-33, track.py) -- only where /root/reference is importable
  oracle    : oracle/tracker.py `sieve_rows` with the oracle's filter restatements (always available)
  native    : rows as a drop-mode camera delivers them (failing rows zeroed) -> `HipTrackFilter().sieve()`
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from watsor_amd.filter.track import HipTrackFilter                    # noqa: E402
from watsor_amd.share import BoundingBox, Detection, DetectionArray   # noqa: E402

CONFIG = {"width": 1280, "height": 720,
          "detect": [{"person": {"area": 0, "confidence": 30, "zones": []}},
                     {"bicycle": {"area": 0, "confidence": 30, "zones": []}},
                     {"car": {"area": 0, "confidence": 30, "zones": []}},
                     {"truck": {"area": 0, "confidence": 30, "zones": []}},
                     {"dog": {"area": 0, "confidence": 30, "zones": []}}]}


def frames_of(seq):
    out = []
    for fr in seq["frames"]:
        arr = DetectionArray()
        k = 0
        for r in fr["rows"]:
            if r[0] <= 0:
                continue
            arr[k] = Detection(label=r[0], confidence=r[2], bounding_box=BoundingBox(*r[3]))
            k += 1
        for i in range(k, 100):
            arr[i].label = 1
        out.append(arr)
    return out


def clone(arr):
    c = DetectionArray()
    import ctypes
    ctypes.memmove(c, arr, ctypes.sizeof(arr))
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--json")
    args = ap.parse_args()
    seqs = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests", "golden", "track.json")))}
    scene = frames_of(seqs["defaults_street"]) + frames_of(seqs["zones_porch"])
    result = {"frames": args.frames, "rows_per_frame": 100,
              "mean_detections_per_frame": sum(sum(1 for d in a if d.confidence > 0) for a in scene) / len(scene)}

    # --- native: the GPU already applied the filters (drop mode); emulate that once, outside the timed region
    from oracle import filters as of
    filters = [of.ConfidenceFilter(CONFIG), of.AreaFilter(CONFIG)]
    dropped = []
    for a in scene:
        c = clone(a)
        for i in range(100):
            if not (c[i].label > 0 and all(f(c[i]) for f in filters)):
                c[i] = Detection()
        dropped.append(c)
    work = [clone(a) for a in dropped]
    native = HipTrackFilter()
    import ctypes
    t0 = time.perf_counter()
    for i in range(args.frames):
        k = i % len(scene)
        ctypes.memmove(work[k], dropped[k], 7200)                 # the detector's write into shared memory
        native.sieve(work[k])
    result["native_us_per_frame"] = (time.perf_counter() - t0) / args.frames * 1e6

    # --- oracle restatement (Python)
    from oracle.tracker import TrackFilter as OracleTrack, sieve_rows
    flt = OracleTrack(filters)
    n = max(50, args.frames // 10)
    t0 = time.perf_counter()
    for i in range(n):
        sieve_rows([flt], [clone(scene[i % len(scene)])[j] for j in range(100)])
    result["oracle_us_per_frame"] = (time.perf_counter() - t0) / n * 1e6

    # --- the reference itself, when importable
    if os.path.isdir("/root/reference/watsor"):
        sys.path.insert(0, "/root/reference")
        from watsor.filter.area import AreaFilter
        from watsor.filter.confidence import ConfidenceFilter
        from watsor.filter.sieve import DetectionSieve
        from watsor.filter.track import TrackFilter
        ref = [TrackFilter([ConfidenceFilter(CONFIG), AreaFilter(CONFIG)])]
        work = [clone(a) for a in scene]
        t0 = time.perf_counter()
        for i in range(n):
            k = i % len(scene)
            ctypes.memmove(work[k], scene[k], 7200)
            dets = [DetectionSieve._clone(d) for d in work[k]]                    # sieve.py:22,44-45
            for f in ref:
                dets, _ = f(dets)                                                  # sieve.py:24-26
            DetectionSieve._copy_to(work[k], dets)                                 # sieve.py:27
        result["reference_us_per_frame"] = (time.perf_counter() - t0) / n * 1e6
        result["speedup_vs_reference"] = result["reference_us_per_frame"] / result["native_us_per_frame"]
    print(json.dumps(result))
    if args.json:
        json.dump(result, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
