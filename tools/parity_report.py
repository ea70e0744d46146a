"""Row-level parity report of the engine against the CPU oracle (GPU tool): per resolution the largest score and box deviations
of the matched rows and every row without a partner with its reason (oracle/compare.py).  The numbers behind the stated box
tolerance and behind "every unmatched row is explained" (DESIGN.md section 4).

usage: python tools/parity_report.py [--robust] [--spread D] [--frames N]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from oracle import compare as cmp                           # noqa: E402
from oracle import detect as odet                           # noqa: E402
from watsor_amd import engine                               # noqa: E402
from watsor_amd.runtime import ROW_DTYPE, HipEngine         # noqa: E402
from watsor_amd.share import DetectionArray                 # noqa: E402
from watsor_amd.synth import spread_channel_scales, synthetic_frame, synthetic_weights   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robust", action="store_true")
    ap.add_argument("--spread", type=float, default=0.0)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    W = synthetic_weights(a.seed)
    if a.spread > 0:
        W = spread_channel_scales(W, a.spread)
    path = os.path.join(tempfile.mkdtemp(), "mi355x.bin")
    engine.save_engine(engine.build_engine(W, robust=a.robust), path)
    oracle = odet.OracleObjectDetector(weights=W)
    eng = HipEngine(path, 0, 8, 1920, 1080)
    print("program %s, channel spread %.1f decades, %d frames per size" % ("robust" if a.robust else "default", a.spread, a.frames))
    try:
        for (w, h) in ((640, 480), (1280, 720), (1920, 1080), (300, 300)):
            agg = dict(ds=0.0, dpx=0, rows=0, missing=0, extra=0, unexplained=0)
            hist = {}
            for i in range(a.frames):
                f = synthetic_frame(w, h, 7000 + 31 * i + w)
                rows = DetectionArray()
                eng.detect_batch([f], [rows])
                got = np.frombuffer(rows, dtype=ROW_DTYPE)
                b, c, s, _, _ = oracle.raw(f)
                ref = odet.rows_as_array(f.shape, b, c, s)
                r = cmp.compare_rows(got, ref, f.shape)
                agg["ds"] = max(agg["ds"], r["max_dscore"]); agg["dpx"] = max(agg["dpx"], r["max_dbox_px"])
                agg["rows"] += len(r["pairs"]); agg["missing"] += len(r["missing"]); agg["extra"] += len(r["extra"])
                agg["unexplained"] += r["unexplained"]
                for p in r["pairs"]:
                    hist[p[4]] = hist.get(p[4], 0) + 1
                for kind, lst in (("missing", r["missing"]), ("extra", r["extra"])):
                    for idx, why in lst:
                        if kind == "missing":
                            desc = "oracle row %d: label %d score %.5f box %s" % (idx, ref["label"][idx], ref["confidence"][idx], list(ref["box"][idx]))
                        else:
                            desc = "gpu row %d: label %d score %.5f box %s" % (idx, got["label"][idx], got["confidence"][idx],
                                                                              [int(got[k][idx]) for k in ("x_min", "y_min", "x_max", "y_max")])
                        print("   %dx%d frame %d %s %s -> %s" % (w, h, i, kind, desc, why or "UNEXPLAINED"))
            print("%4dx%-4d rows matched %4d  max |dscore| %.2e  max |dbox| %d px (stated tolerance %d px; histogram %s)  "
                  "missing %d extra %d unexplained %d" % (w, h, agg["rows"], agg["ds"], agg["dpx"], cmp.box_tolerance_px(w, h),
                                                          dict(sorted(hist.items())), agg["missing"], agg["extra"], agg["unexplained"]), flush=True)
    finally:
        eng.close()


if __name__ == "__main__":
    main()
