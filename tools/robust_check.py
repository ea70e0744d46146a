"""Score error of the `-p 16` programs (default / robust) against the CPU oracle on weights of a given channel spread (GPU tool).
usage: python tools/robust_check.py [decades ...]   (0 = the seeded He-initialised weights as they are)"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
os.environ.setdefault("WATSOR_HIP_DEV", "1")

import parity_utils as pu                                   # noqa: E402
from oracle import detect as odet                           # noqa: E402
from watsor_amd import engine                               # noqa: E402
from watsor_amd.runtime import ROW_DTYPE, HipEngine         # noqa: E402
from watsor_amd.share import DetectionArray                # noqa: E402
from watsor_amd.synth import spread_channel_scales, synthetic_frame, synthetic_weights   # noqa: E402


def worst(path, oracle_rows, frames):
    eng = HipEngine(path, 0, 8, 640, 480)
    w, n = 0.0, 0
    try:
        for f, ref in zip(frames, oracle_rows):
            rows = DetectionArray()
            eng.detect_batch([f], [rows])
            got = np.frombuffer(rows, dtype=ROW_DTYPE)
            pairs, missing = pu.match_rows(got, ref, min_score=0.1)
            w = max(w, max((abs(p[3]) for p in pairs), default=9.99))      # (9.99: no row found a partner)
            n += len(pairs)
    finally:
        eng.close()
    return w, n


def main():
    decades = [float(a) for a in sys.argv[1:]] or [0.0, 1.0, 1.5]
    nfr = int(os.environ.get("NFRAMES", "3"))
    frames = [synthetic_frame(640, 480, 5000 + i) for i in range(nfr)]
    base = synthetic_weights(1234)
    tmp = tempfile.mkdtemp()
    for d in decades:
        W = spread_channel_scales(base, d) if d > 0 else base
        oracle = odet.OracleObjectDetector(weights=W)
        t0 = time.time()
        refs = []
        for f in frames:
            b, c, s, _, _ = oracle.raw(f)
            refs.append(odet.rows_as_array(f.shape, b, c, s))
        line = "spread %.1f decades (measured %.2f, oracle %.0f s):" % (d, engine.channel_spread_decades(W), time.time() - t0)
        for name, kw in (("default", {}), ("robust", {"robust": True, "float_form_upto": int(os.environ.get("WZ_FLOAT_UPTO", "9")),
                                                    "conv1_split": os.environ.get("WZ_CONV1_SPLIT", "1") != "0"})):
            path = os.path.join(tmp, "%s_%s.bin" % (name, d))
            engine.save_engine(engine.build_engine(W, 16, **kw), path)
            e, n = worst(path, refs, frames)
            line += "  %s %.2e" % (name, e)
        print(line + "  (%d rows)" % n, flush=True)


if __name__ == "__main__":
    main()
