"""A/B of how page-locked HOST frames reach the resize kernel (GPU tool, development library):
   WZ_PRE_ROWS=0            per-pixel resize kernel, frames staged by one DMA each (round 3)
   WZ_PRE_ROWS=1            row-staged resize kernel, frames still staged
   WZ_HOST_READ=1           row-staged kernel reads every frame IN PLACE over PCIe (no staging copy)
   WZ_HOST_READ=2           ... only frames whose vertical down-scale skips rows (>= 2x), the others staged
For each: frames/s with 4 lanes in flight and p50 of a lone step, on 640x480 frames in HBM (the headline workload: the resize
kernel's own cost), 640x480 / 1920x1080 RGB24 / 1920x1080 NV12 frames in page-locked host memory, and a 16-camera mix.  The rows of
every mode are compared bit for bit with mode 0's.

    python tools/host_read_ab.py [--quick]
"""
import os
import sys
import time

os.environ["WATSOR_HIP_DEV"] = "1"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from watsor_amd import engine as eb                                   # noqa: E402
from watsor_amd.runtime import FMT_NV12, HipEngine                    # noqa: E402
from watsor_amd.synth import synthetic_frame, synthetic_weights      # noqa: E402

MODES = [("per-pixel kernel, staged", dict(WZ_PRE_ROWS="0", WZ_HOST_READ="0")),
         ("row kernel, staged", dict(WZ_PRE_ROWS="1", WZ_HOST_READ="0")),
         ("row kernel, in place", dict(WZ_HOST_READ="1")),
         ("row kernel, in place if >= 2x", dict(WZ_HOST_READ="2")),
         ("defaults (per-pixel kernel for staged batches, row kernel in place for frames >= 2x)", dict())]
# (tried and removed, round 4: the staged frames of a batch as ONE hipMemcpyBatchAsync -- 32.6 k against 32.7 k frames/s at 640x480,
# the runtime issues the same copies one by one; profiles/r04_host_read_ab.txt)


def throughput(eng, submit, n, steps, warm=16):
    lanes = eng.num_slots
    for s in range(warm):
        submit(s % lanes, s)
    eng.sync()
    t0 = time.perf_counter()
    for s in range(steps):
        submit(s % lanes, s)
    eng.sync()
    dt = time.perf_counter() - t0
    lat = []
    for s in range(40):
        t1 = time.perf_counter()
        submit(0, s)
        eng.wait(0)
        lat.append((time.perf_counter() - t1) * 1e3)
    return steps * n / dt, float(np.median(lat))


def main():
    quick = "--quick" in sys.argv
    path = "/tmp/wz_hostread/mi355x.bin"
    os.makedirs(os.path.dirname(path), exist_ok=True)
    eb.save_engine(eb.build_engine(synthetic_weights(1234)), path)
    small = np.stack([synthetic_frame(640, 480, 100 + i) for i in range(16)])
    big = np.stack([synthetic_frame(1920, 1080, 200 + i) for i in range(8)])
    nv = np.zeros((8, 1620, 1920), np.uint8)
    for i in range(8):
        nv[i] = big[i].reshape(-1)[:nv[i].size].reshape(1620, 1920)
    ref_rows = {}
    for label, env in (MODES[-1:] if "--defaults" in sys.argv else MODES):
        for k in ("WZ_PRE_ROWS", "WZ_HOST_READ", "WZ_BATCH_COPY"):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = HipEngine(path, 0, 16, 1920, 1080, dev=True)
        try:
            res = {}
            d = [eng.upload(small[i]) for i in range(8)]
            res["hbm 640x480 b8"] = throughput(eng, lambda lane, s: eng.submit_device(lane, d, [640] * 8, [480] * 8), 8, 200 if quick else 400)
            for arena in (small, big, nv):
                eng.host_register(arena)
            v = [[small[b * 8 + i] for i in range(8)] for b in range(2)]
            res["host 640x480 b8"] = throughput(eng, lambda lane, s: eng.submit_host(lane, v[s % 2]), 8, 200 if quick else 400)
            vb = [big[i] for i in range(8)]
            res["host 1080p rgb24 b8"] = throughput(eng, lambda lane, s: eng.submit_host(lane, vb), 8, 40 if quick else 80)
            vb4 = [[big[(b * 4 + i) % 8] for i in range(4)] for b in range(2)]
            res["host 1080p rgb24 b4"] = throughput(eng, lambda lane, s: eng.submit_host(lane, vb4[s % 2]), 4, 60 if quick else 120)
            vn = [nv[i] for i in range(8)]
            res["host 1080p nv12 b8"] = throughput(eng, lambda lane, s: eng.submit_host(lane, vn, formats=[FMT_NV12] * 8), 8, 40 if quick else 80)
            mix = []
            for c in range(8):
                mix += [small[c], big[c]]
            res["host 16 mixed b16"] = throughput(eng, lambda lane, s: eng.submit_host(lane, mix), 16, 30 if quick else 60)
            # rows of this mode on one batch of each kind, against mode 0
            same = []
            for name, frames, fmts in (("640x480", v[0], None), ("1080p", vb, None), ("nv12", vn, [FMT_NV12] * 8), ("mixed", mix, None)):
                eng.submit_host(0, frames, formats=fmts)
                eng.wait(0)
                rows = eng.slot_rows(0, len(frames)).copy().tobytes()
                if name in ref_rows:
                    same.append("%s %s" % (name, "identical" if rows == ref_rows[name] else "DIFFERENT"))
                else:
                    ref_rows[name] = rows
            eng.sync()
            for arena in (small, big, nv):
                eng.host_unregister(arena)
        finally:
            eng.close()
        print("== %s" % label)
        for k, (fps, p50) in res.items():
            print("   %-22s %9.0f frames/s   p50 %.3f ms" % (k, fps, p50))
        if same:
            print("   rows vs mode 0: " + ", ".join(same))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
