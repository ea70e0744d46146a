"""PCIe-inclusive throughput: frames handed over as HOST pointers (the reference's boundary: a view of the shared
FrameBuffer mmap, watsor/detection/detector.py:104-106), batch = 8, 640x480 (and 1920x1080), pageable vs page-locked.

    python tools/host_path_bench.py  [--out profiles/xxx.json]
"""
import json
import os
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from watsor_amd import engine as eb                                   # noqa: E402
from watsor_amd.runtime import HipEngine                              # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights      # noqa: E402

BATCH = 8


def run(eng, frames, n_steps):
    lanes = eng.num_slots
    ring = len(frames) // BATCH
    for s in range(2 * lanes):
        eng.submit_host(s % lanes, frames[(s % ring) * BATCH:(s % ring + 1) * BATCH])
    eng.sync()
    t0 = time.perf_counter()
    for s in range(n_steps):
        eng.submit_host(s % lanes, frames[(s % ring) * BATCH:(s % ring + 1) * BATCH])
    eng.sync()
    return n_steps * BATCH / (time.perf_counter() - t0)


def main():
    out = {}
    path = "/tmp/wz_hostbench/mi355x.bin"
    os.makedirs(os.path.dirname(path), exist_ok=True)
    eb.save_engine(eb.build_engine(synthetic_weights(1234)), path)
    for (w, h) in [(640, 480), (1920, 1080)]:
        eng = HipEngine(path, 0, BATCH, w, h)
        ring = 4
        arena = np.empty((ring * BATCH, h, w, 3), np.uint8)           # stands in for a FrameBuffer arena
        for i in range(ring * BATCH):
            arena[i] = synthetic_frame(w, h, 1234 + i)
        frames = [arena[i] for i in range(ring * BATCH)]
        pageable = run(eng, frames, 100)
        eng.host_register(arena)
        pinned = run(eng, frames, 200)
        eng.host_unregister(arena)
        eng.close()
        mb = w * h * 3 / 1e6
        out["%dx%d" % (w, h)] = dict(pageable_fps=round(pageable, 1), registered_fps=round(pinned, 1),
                                     registered_h2d_gbs=round(pinned * mb / 1e3, 2), frame_mb=round(mb, 3))
        print("%dx%d: pageable %.0f frames/s, page-locked %.0f frames/s (%.1f GB/s of H2D)" % (w, h, pageable, pinned, pinned * mb / 1e3))
    if "--out" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--out") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
