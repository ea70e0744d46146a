"""How many leading blocks carry split operands (`hp_upto`): score error and throughput of the headline workload for each choice.

    python tools/hp_sweep.py [--from 6] [--to 14]      (GPU box)
Prints, per program: max |dscore| vs the oracle over 16 frames (two batches of the benchmark's frames), frames/s with four lanes in
flight, p50 of a synchronous step.  hp_upto = -1 is the --plain-fp16 program; 12 is the default of `-p 16`.
"""
import argparse
import os
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.compare import match_rows                                 # noqa: E402
from oracle.detect import OracleObjectDetector, rows_as_array         # noqa: E402
from watsor_amd import engine as eb                                   # noqa: E402
from watsor_amd.runtime import HipEngine                              # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--from", dest="lo", type=int, default=6)
ap.add_argument("--to", dest="hi", type=int, default=14)
args = ap.parse_args()
B, W, H = 8, 640, 480
weights = synthetic_weights(1234)
frames = [synthetic_frame(W, H, 1234 + i) for i in range(2 * B)]
oracle = OracleObjectDetector(weights=weights)
refs = []
for f in frames:
    b, c, s, _, _ = oracle.raw(f)
    refs.append(rows_as_array(f.shape, b, c, s))
for hp in [-1] + list(range(args.lo, args.hi + 1)):
    path = "/tmp/wz_hp_sweep/mi355x.bin"
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        eb.save_engine(eb.build_engine(weights, hp_upto=hp), path)
    except Exception as exc:                                           # (a block the split kernel has no shape for)
        print("hp_upto %3d: not buildable: %s" % (hp, exc))
        continue
    eng = HipEngine(path, 0, B, W, H)
    try:
        d = [eng.upload(f) for f in frames]
        worst = 0.0
        for k in range(2):
            eng.submit_device(0, d[k * B:(k + 1) * B], [W] * B, [H] * B)
            eng.wait(0)
            got = eng.slot_rows(0, B).copy()
            for i in range(B):
                pairs, _ = match_rows(got[i], refs[k * B + i], min_score=0.0)
                worst = max([worst] + [abs(p[3]) for p in pairs])
        lanes = eng.num_slots
        for s_ in range(40):
            eng.submit_device(s_ % lanes, d[:B], [W] * B, [H] * B)
        eng.sync()
        rates = []
        for _ in range(5):
            t0 = time.perf_counter()
            for s_ in range(200):
                eng.submit_device(s_ % lanes, d[:B], [W] * B, [H] * B)
            eng.sync()
            rates.append(200 * B / (time.perf_counter() - t0))
        lat = []
        for s_ in range(100):
            t1 = time.perf_counter()
            eng.submit_device(0, d[:B], [W] * B, [H] * B)
            eng.wait(0)
            lat.append((time.perf_counter() - t1) * 1e3)
        print("hp_upto %3d: split blocks %2d   max|dscore| %.2e   %6.0f frames/s   p50 %.4f ms"
              % (hp, eng.hp_blocks, worst, float(np.median(rates)), float(np.median(lat))), flush=True)
    finally:
        eng.close()
