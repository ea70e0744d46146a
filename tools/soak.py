"""Soak: the same batches over all lanes for N seconds, every batch's rows compared bit for bit with the first result for those frames
(the pipeline is deterministic: fixed summation orders everywhere).  Catches races that a single pass of the parity tests can miss.
usage: python tools/soak.py [seconds] [--robust] [--small]     (--small: batches of 1 .. 5 frames -- launches that leave most of the chip empty)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from watsor_amd import engine as eb                                   # noqa: E402
from watsor_amd.runtime import HipEngine                              # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights      # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 30.0
robust = "--robust" in sys.argv
small = "--small" in sys.argv
path = "/tmp/wz_soak/mi355x.bin"
os.makedirs(os.path.dirname(path), exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234), robust=robust), path)
eng = HipEngine(path, 0, 8, 1280, 720)
sizes = [(640, 480), (1280, 720), (320, 240), (640, 360)]
sets = []
for k in range(5):                                  # five different batches (sizes 8, 7, 6, 5, 4; mixed resolutions), rotating over the lanes
    n = k + 1 if small else 8 - k
    fr = [synthetic_frame(*sizes[(k + i) % 4], 9000 + 10 * k + i) for i in range(n)]
    sets.append(([eng.upload(f) for f in fr], [f.shape[1] for f in fr], [f.shape[0] for f in fr]))
lanes = eng.num_slots
ref, inflight, steps, bad = {}, {}, 0, 0
t0 = time.time()
while time.time() - t0 < secs:
    lane = steps % lanes
    if lane in inflight:
        k = inflight.pop(lane)
        eng.wait(lane)
        rows = eng.slot_rows(lane, len(sets[k][0])).copy().tobytes()
        if k not in ref:
            ref[k] = rows
        elif rows != ref[k]:
            bad += 1
    k = steps % len(sets)
    d, w, h = sets[k]
    eng.submit_device(lane, d, w, h)
    inflight[lane] = k
    steps += 1
eng.sync()
print("%s program%s: %d steps in %.0f s, %d batches differed from their first result" % ("robust" if robust else "default", ", batches of 1 .. 5" if small else "", steps, time.time() - t0, bad))
eng.close()
sys.exit(1 if bad else 0)
