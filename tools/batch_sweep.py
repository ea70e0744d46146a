"""Batch sweep (BASELINE configs 4-5 put 4 .. 16 cameras on a GPU; saturation runs go further): throughput and
per-kernel-class roofline fractions at batch 8 / 16 / 32 / 64, one lane vs four.

    python tools/batch_sweep.py [--out profiles/rNN_batch_sweep.json]

Per batch size: frames/s with 4 lanes and with WZ_LANES=1, p50 of a synchronous step, and for every kernel class the
time per step, the achieved GB/s / TFLOP/s under SURVEY 8(d)'s per-layer rule and the fraction of the roofline
(bench.py's aggregate_stages; launches timed 4 per bracket).
"""
import argparse
import json
import os
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(batch, lanes):
    os.environ["WZ_LANES"] = str(lanes)
    import bench
    from watsor_amd import engine as eb
    from watsor_amd.runtime import HipEngine
    from watsor_amd.synth import synthetic_frame, synthetic_weights
    default_program = os.environ.get("WZ_SWEEP_DEFAULT_PROGRAM", "0") != "0"    # (the headline's robust program unless asked otherwise)
    path = "/tmp/wz_sweep/mi355x_%s.bin" % ("default" if default_program else "robust")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if not os.path.isfile(path):
        eb.save_engine(eb.build_engine(synthetic_weights(1234), **({} if default_program else bench.HEADLINE_PROGRAM)), path)
    eng = HipEngine(path, 0, batch, 640, 480)
    d = [eng.upload(synthetic_frame(640, 480, 1234 + i % 16)) for i in range(batch)]
    ws, hs = [640] * batch, [480] * batch
    r = bench.throughput(eng, lambda lane, s: eng.submit_device(lane, d, ws, hs), batch, steps=max(40, 640 // batch), warm=8)
    out = dict(batch=batch, lanes=eng.num_slots, frames_per_s=r["value"], ms_per_step=r["ms_per_step"], p50_ms=r["p50_ms"])
    if lanes == 1:
        stages = eng.profile_device(d, ws, hs, reps=6, inner=4)
        table, ov = bench.aggregate_stages(stages, eng.ops(), batch, 640 * 480 * 3, eng.input_size, eng.hp_blocks, 4)
        out["kernels"] = [dict(kernel=t["kernel"], launches=t["launches"], us_per_step=round(t["ms_per_step"] * 1e3, 2),
                               bound=t["bound"], gbs=round(t["gbs"], 1), tflops=round(t["tflops"], 1),
                               frac=round(t["t_roof_frac"], 4)) for t in table]
    eng.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--one", nargs=2, type=int, default=None)
    ap.add_argument("--batches", default="8,16,32,64")
    args = ap.parse_args()
    if args.one:                       # child: one engine per process (WZ_LANES is read when the engine is created)
        print("RESULT " + json.dumps(one(*args.one)))
        sys.exit(0)
    res = []
    for b in [int(x) for x in args.batches.split(",")]:
        for lanes in (4, 1):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(b), str(lanes)],
                               capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(p.stderr[-2000:])
                continue
            r = json.loads(line[0][7:])
            res.append(r)
            print("batch %3d lanes %d: %8.0f frames/s  %.4f ms/step  p50 %.4f ms" % (r["batch"], r["lanes"], r["frames_per_s"],
                                                                               r["ms_per_step"], r["p50_ms"]))
            for k in r.get("kernels", [])[:8]:
                print("      %-24s x%-3d %8.2f us  %-4s %8.1f GB/s %8.1f TF  frac %.3f" % (
                    k["kernel"], k["launches"], k["us_per_step"], k["bound"], k["gbs"], k["tflops"], k["frac"]))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
