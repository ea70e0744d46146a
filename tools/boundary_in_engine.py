"""Per-boundary gaps of a LONE batch inside the engine (in-kernel stamps, `make -C watsor_amd/csrc stamps`): earliest entry of launch k + 1 minus
latest exit of launch k, median over the steps, one lane -- to hold against tools/micro/boundary.hip (the same measure on trivial kernels).

    WZ_LANES=1 WZ_GRAPH=1 python tools/boundary_in_engine.py     (captured graph)      WZ_LANES=1 WZ_GRAPH=0 ...  (kernel by kernel)
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("WATSOR_HIP_DEV_LIBRARY", os.path.join(ROOT, "watsor_amd", "libwatsor_hip_stamps.so"))
os.environ.setdefault("WATSOR_HIP_DEV", "1")
os.environ.setdefault("WZ_LANES", "1")
import numpy as np                         # noqa: E402
import lane_overlap as lo                  # noqa: E402
from watsor_amd import engine as eb        # noqa: E402
from watsor_amd.synth import synthetic_weights   # noqa: E402

path = "/tmp/wz_boundary/mi355x.bin"
os.makedirs(os.path.dirname(path), exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234), robust="--default-program" not in sys.argv), path)
batch = 8
iv, launches, fps, ms, lanes = lo.collect(path, 200, 20, batch)
by_step = {}
for lane, step, k, t0, t1 in iv:
    by_step.setdefault(step, {})[k] = (t0, t1)
steps = sorted(by_step)[5:-5]
nl = max(len(by_step[s]) for s in steps)
gaps = [[] for _ in range(nl)]
durs = [[] for _ in range(nl)]
for s in steps:
    d = by_step[s]
    for k in range(nl):
        if k in d:
            durs[k].append((d[k][1] - d[k][0]) * lo.TICK_US)
            if k > 0 and k - 1 in d:
                gaps[k].append((d[k][0] - d[k - 1][1]) * lo.TICK_US)
print("# lone batches of %d, one lane, WZ_GRAPH=%s: %.0f frames/s, %.4f ms per step; launch k: duration (first entry .. last exit) and the gap in FRONT of it, us (medians over %d steps)"
      % (batch, os.environ.get("WZ_GRAPH", "(unset: kernel by kernel when the other lanes are idle)"), fps, ms, len(steps)))
tot_g = tot_d = 0.0
for k in range(nl):
    name = launches[k]["kernel"][:58] if k < len(launches) else "?"
    g = float(np.median(gaps[k])) if gaps[k] else 0.0
    d = float(np.median(durs[k])) if durs[k] else 0.0
    tot_g += g
    tot_d += d
    print("%2d %-58s wgs %5d  dur %6.2f  gap %5.2f" % (k, name, launches[k]["workgroups"] if k < len(launches) else 0, d, g))
print("sum of durations %.1f us, sum of gaps %.1f us over %d boundaries = %.2f us each" % (tot_d, tot_g, nl - 1, tot_g / max(1, nl - 1)))
