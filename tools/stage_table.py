"""Per-kernel HIP-event times of one batch (wz_profile_device) for either engine.

    python tools/stage_table.py [--precision 16|32] [--batch 8] [--unfused]
"""
import argparse
import os
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
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
ap.add_argument("--plain-fp16", action="store_true", help="the -p 16 program without split-operand blocks")
ap.add_argument("--robust", action="store_true", help="the robust -p 16 program (all 17 blocks split, float-form chunk buffer up to block WZ_FLOAT_UPTO, default 12)")
ap.add_argument("--tap-conv", action="store_true", help="block 13's expand conv as a launch of its own (the program before the block stored it itself)")
ap.add_argument("--inner", type=int, default=1, help="launches per bracket (wz_profile_stages): > 1 = a launch incl. its in-stream boundary")
ap.add_argument("--only", default="", help="print only the stages whose name contains this")
ap.add_argument("--throughput", action="store_true", help="also: frames/s with the lanes in flight and p50 of synchronous steps")
args = ap.parse_args()
path = "/tmp/wz_stage_table/mi355x.bin"
os.makedirs(os.path.dirname(path), exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234), precision=args.precision, fuse=not args.unfused,
                               hp_upto=-1 if args.plain_fp16 else None, robust=args.robust, tap_in_block=not args.tap_conv,
                               float_form_upto=int(os.environ.get("WZ_FLOAT_UPTO", "9")),
                               conv1_split=(os.environ.get("WZ_CONV1_SPLIT", "1") != "0") if args.robust else None), path)
eng = HipEngine(path, 0, args.batch, 640, 480)
d = [eng.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(args.batch)]
for _ in range(20):
    eng.submit_device(0, d, [640] * args.batch, [480] * args.batch)
eng.sync()
stages = eng.profile_device(d, [640] * args.batch, [480] * args.batch, reps=20, inner=args.inner)
_c = sorted(ms for n, ms in stages if n == "(empty)" or n.endswith("#splitk_reduce"))
_near = [ms for ms in _c if ms <= _c[0] + 1e-3]
ov = _near[len(_near) // 2]   # median of the empty brackets (see bench.py: empty_bracket_ms)
tot = 0.0
for n, ms in stages:
    v = (ms - ov) * 1e3 / (1 if (n.startswith("post/") or n in ("(empty)", "h2d_descriptors")) else args.inner)
    if v > 0.3:
        tot += v
        if args.only in n:
            print("%-72s %8.2f" % (n, v))
print("sum %.1f us (event bracket %.2f us subtracted per stage, %d launch(es) per bracket)" % (tot, ov * 1e3, args.inner))
if args.throughput:
    import time
    import numpy as np
    lanes, B = eng.num_slots, args.batch
    for s_ in range(40):
        eng.submit_device(s_ % lanes, d, [640] * B, [480] * B)
    eng.sync()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for s_ in range(200):
            eng.submit_device(s_ % lanes, d, [640] * B, [480] * B)
        eng.sync()
        best.append(200 * B / (time.perf_counter() - t0))
    lat = []
    for s_ in range(100):
        t1 = time.perf_counter()
        eng.submit_device(0, d, [640] * B, [480] * B)
        eng.wait(0)
        lat.append((time.perf_counter() - t1) * 1e3)
    print("throughput %.0f frames/s (median of 5 x 200 steps; min %.0f max %.0f), p50 %.4f ms" % (np.median(best), min(best), max(best), np.median(lat)))
