"""Per-kernel HIP-event times of one batch (wz_profile_device) for either engine.

    python tools/stage_table.py [--precision 16|32] [--batch 8] [--unfused]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from watsor_amd import engine as eb                                   # noqa: E402
from watsor_amd.runtime import HipEngine                              # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", type=int, default=16)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--unfused", action="store_true")
args = ap.parse_args()
path = "/tmp/wz_stage_table/mi355x.bin"
os.makedirs(os.path.dirname(path), exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234), precision=args.precision, fuse=not args.unfused), path)
eng = HipEngine(path, 0, args.batch, 640, 480)
d = [eng.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(args.batch)]
for _ in range(20):
    eng.submit_device(0, d, [640] * args.batch, [480] * args.batch)
eng.sync()
stages = eng.profile_device(d, [640] * args.batch, [480] * args.batch, reps=20)
_c = sorted(ms for n, ms in stages if n == "(empty)" or n.endswith("#splitk_reduce"))
_near = [ms for ms in _c if ms <= _c[0] + 1e-3]
ov = _near[len(_near) // 2]   # median of the empty brackets (see bench.py: empty_bracket_ms)
tot = 0.0
for n, ms in stages:
    v = (ms - ov) * 1e3
    if v > 0.3:
        tot += v
        print("%-72s %8.2f" % (n, v))
print("sum %.1f us (event bracket %.2f us subtracted per stage)" % (tot, ov * 1e3))
