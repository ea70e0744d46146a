"""Throughput / latency / roofline bench of the MI355X detection hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (resize+normalise -> SSD-MobileNet-v2 -> decode -> NMS -> 100
Detection rows per frame, D2H of the rows included) over one batch of synthetic frames that are
already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: one synthetic 640x480 RGB
stream, batch = 8 frames, 1 x MI355X.  For N > 1 every rank is an independent replica with its own
camera (cameras are the shard; no data-path collective -- SURVEY.md 8e), value = frames of all ranks
/ max-over-ranks time, scaling "weak".

Rank 0 prints ONE JSON line (contract in the task description) extended with
  "p50_ms"      : median per-step latency of synchronous steps (the other half of BASELINE's metric)
  "roofline"    : dominant kernel of the step, timed live with HIP events on the engine's own stream
  "cpu_baseline": the oracle (CPU restatement of the reference's TF detector) on this host's cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense fp16/bf16 MFMA
WIDTH, HEIGHT, BATCH = 640, 480, 8


def kernel_class(op):
    from watsor_amd import arch
    if op["kind"] == arch.OP_STEM:
        return "wz_k_stem"
    if op["kind"] == arch.OP_DW:
        return "wz_k_dw"
    if op["kind"] == arch.OP_MBCONV:
        return "wz_k_mbconv"
    return "wz_k_conv<%d>" % op["ksize"]


def algorithmic_cost(op, n):
    """(flops, bytes) of one launch per the per-layer rule of SURVEY.md 8(d): every tensor once.
    A fused inverted-residual block is priced as the layers it computes (expand, depthwise, project), i.e.
    the bytes a layer-by-layer execution moves; `fused_min_bytes` below is what the fused launch has to move."""
    from watsor_amd import arch
    M = n * op["hout"] * op["wout"]
    if op["kind"] == arch.OP_MBCONV:
        Min, cin, cmid, cout = n * op["hin"] * op["win"], op["cin"], op["cmid"], op["cout"]
        fl, by = 0.0, 0.0
        if cin == 3:                                                 # stem conv folded in (3x3 s2 on the 4-channel input)
            fl += 2.0 * Min * 32 * 27
            by += 2.0 * (n * (2 * op["hin"]) * (2 * op["win"]) * 4 + Min * 32) + 27 * 32 * 4
        elif cin != cmid:                                            # 1x1 expand
            fl += 2.0 * Min * cin * cmid
            by += 2.0 * (Min * cin + cin * cmid + Min * cmid)
        fl += 2.0 * M * cmid * 9                                     # depthwise 3x3
        by += 2.0 * (Min * cmid + 9 * cmid + M * cmid)
        fl += 2.0 * M * cmid * cout                                  # 1x1 project
        by += 2.0 * (M * cmid + cmid * cout + M * cout)
        return fl, by
    if op["kind"] == arch.OP_DW:
        c = op["cin"]
        return 2.0 * M * c * 9, 2.0 * (n * op["hin"] * op["win"] * c + 9 * c + M * c)
    if op["kind"] == arch.OP_STEM:
        return 2.0 * M * 32 * 27, 2.0 * (n * op["hin"] * op["win"] * 4 + M * 32) + 27 * 32 * 4
    K = op["ksize"] ** 2 * op["cin"]
    N = op["cout"]
    out_bytes = 4.0 if op["name"].startswith("BoxPredictor") else 2.0     # head outputs are fp32
    return 2.0 * M * N * K, 2.0 * (n * op["hin"] * op["win"] * op["cin"] + K * N) + out_bytes * M * N


def empty_bracket_ms(stages):
    """Cost of an event bracket with no kernel in it: the median of the brackets that are empty ("(empty)" and the
    unused split-K slots), i.e. of those within 1 us of the shortest one (back-to-back empty brackets read ~0.5 us
    shorter than isolated ones, so the minimum itself would inflate every other stage)."""
    cands = sorted([ms for name, ms in stages if name == "(empty)" or name.endswith("#splitk_reduce")])
    near = [ms for ms in cands if ms <= cands[0] + 1e-3]
    return near[len(near) // 2]


def roofline_from_stages(stages, ops, n, frame_bytes, size):
    """Aggregate event-bracketed stage times by kernel; return (roofline dict of the dominant kernel, table)."""
    from watsor_amd import arch
    by_name = {o["name"]: o for o in ops}
    # A hipEventRecord pair with nothing in between reads ~5 us on this stack (the record itself is a
    # barrier packet).  Brackets of unused split-K slots are exactly that: calibrate on them and
    # subtract, so a stage time is the kernel's own duration as rocprofv3's kernel trace reports it.
    overhead = empty_bracket_ms(stages)
    agg = {}
    for name, ms in stages:
        ms = max(ms - overhead, 0.0)
        if name.endswith("#splitk_reduce"):
            k, fl, by = "wz_k_splitk_reduce", 0.0, 0.0
            if ms < 5e-4:
                continue
        elif name in ("heads#small_convs", "heads#big_convs"):   # the SSD heads' shared launches: their flops / bytes are
            k, fl, by = "wz_k_conv<3>", 0.0, 0.0            # counted in their own (empty) op slots below
            if ms < 5e-4:
                continue
        elif name in by_name:
            o = by_name[name]
            k = kernel_class(o)
            fl, by = algorithmic_cost(o, n)
            if ms < 5e-4 and o["kind"] == arch.OP_CONV:     # deferred into the shared launch: work yes, launch no
                a = agg.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, min_bytes=0.0))
                a["flops"] += fl; a["bytes"] += by; a["min_bytes"] += by
                continue
        elif name == "preprocess":
            k, fl, by = "wz_k_preprocess", 0.0, float(n * (frame_bytes + size * size * 4 * 2))
        elif name.startswith("post/"):
            k, fl, by = "wz_k_" + name.split("/")[1], 0.0, 0.0
        else:
            continue
        a = agg.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, min_bytes=0.0))
        a["ms"] += ms; a["flops"] += fl; a["bytes"] += by; a["launches"] += 1
        if name in by_name and by_name[name]["kind"] == arch.OP_MBCONV:   # block input + weights + output, once each
            o = by_name[name]
            cin, cmid, cout = o["cin"], o["cmid"], o["cout"]
            inp = n * (2 * o["hin"]) * (2 * o["win"]) * 4 if cin == 3 else n * o["hin"] * o["win"] * cin
            a["min_bytes"] += 2.0 * (inp + (cin * cmid if cin != cmid else 0) + 9 * cmid +
                                     cmid * cout + n * o["hout"] * o["wout"] * cout)
        else:
            a["min_bytes"] += by
    table = []
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        t = a["ms"] * 1e-3
        t_hbm = a["bytes"] / (HBM_PEAK_GBS * 1e9)
        t_mfma = a["flops"] / (MFMA_PEAK_TFLOPS * 1e12)
        table.append(dict(kernel=k, launches=a["launches"], ms_per_step=a["ms"], avg_us=a["ms"] * 1e3 / a["launches"],
                          gbs=a["bytes"] / t / 1e9 if t > 0 else 0.0, tflops=a["flops"] / t / 1e12 if t > 0 else 0.0,
                          t_roof_frac=max(t_hbm, t_mfma) / t if t > 0 else 0.0,
                          bound="mfma" if t_mfma > t_hbm else "hbm",
                          bytes_per_launch=a["bytes"] / a["launches"], flops_per_launch=a["flops"] / a["launches"],
                          min_bytes_per_launch=a["min_bytes"] / a["launches"],
                          min_gbs=a["min_bytes"] / t / 1e9 if t > 0 else 0.0))
    dom = table[0]
    if dom["bound"] == "hbm":
        ach, peak, unit = dom["gbs"], HBM_PEAK_GBS, "GB/s"
    else:
        ach, peak, unit = dom["tflops"], MFMA_PEAK_TFLOPS, "TFLOP/s"
    roof = dict(kernel=dom["kernel"], event_overhead_us=round(overhead * 1e3, 3), bound=dom["bound"], achieved=round(ach, 3), peak=peak, unit=unit,
                frac=round(ach / peak, 5), traffic=pmc_traffic(dom["kernel"]), avg_launch_us=round(dom["avg_us"], 3),
                launches_per_step=dom["launches"], time_frac_of_roofline=round(dom["t_roof_frac"], 5),
                algorithmic_bytes_per_launch=dom["bytes_per_launch"], algorithmic_flops_per_launch=dom["flops_per_launch"],
                # what the launch has to move when intermediate tensors stay on chip (= algorithmic for unfused kernels)
                fused_min_bytes_per_launch=dom["min_bytes_per_launch"],
                frac_of_peak_on_fused_min_bytes=round(dom["min_gbs"] / HBM_PEAK_GBS, 5))
    rp = rocprof_avg_us(dom["kernel"])
    if rp:   # the same fraction at the duration the committed rocprofv3 trace reports for this kernel
        roof["avg_launch_us_rocprof"] = rp
        roof["frac_at_rocprof_duration"] = round(roof["frac"] * dom["avg_us"] / rp, 5)
    return roof, table


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/pmc_traffic.json, written by tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane reads on gfx950).  None when absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(path)).get(kernel)
        if not t or t["fetch_bytes_corrected"] is None or t["write_bytes"] is None:
            return None
        return dict(bytes=round(t["fetch_bytes_corrected"] + t["write_bytes"]), fetch_bytes_raw=round(t["fetch_bytes_raw"]),
                    fetch_bytes_corrected=round(t["fetch_bytes_corrected"]), write_bytes=round(t["write_bytes"]),
                    source="profiles/pmc_traffic.json (rocprofv3 --pmc, mean per launch)")
    except (OSError, ValueError, KeyError):
        return None


def rocprof_avg_us(kernel):
    """Average per-dispatch duration of `kernel` in the committed rocprofv3 kernel trace of this command
    (profiles/rocprof_kernel_avg.json).  rocprofv3 counts dispatch + teardown into a duration, the HIP-event
    bracket minus the empty-bracket calibration does not: the two differ by 2-3 us per launch."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "rocprof_kernel_avg.json"))).get(kernel, {}).get("avg_us")
    except (OSError, ValueError):
        return None


def fp32_engine_leg(weights, frames, rank):
    """Throughput of the `-p 32` engine (the one that meets the 1e-3 score tolerance) on the same workload."""
    from watsor_amd import engine as builder
    from watsor_amd.runtime import HipEngine
    d = "/tmp/wz_bench32_%d_%d" % (os.getpid(), rank)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "mi355x.bin")
    builder.save_engine(builder.build_engine(weights, precision=32), path)
    eng = HipEngine(path, int(os.environ.get("LOCAL_RANK", "0")), BATCH, WIDTH, HEIGHT)
    try:
        dfr = [eng.upload(f) for f in frames[:BATCH]]
        ws, hs = [WIDTH] * BATCH, [HEIGHT] * BATCH
        for s in range(2 * eng.num_slots):
            eng.submit_device(s % eng.num_slots, dfr, ws, hs)
        eng.sync()
        n = 60
        t0 = time.perf_counter()
        for s in range(n):
            eng.submit_device(s % eng.num_slots, dfr, ws, hs)
        eng.sync()
        dt = time.perf_counter() - t0
    finally:
        eng.close()
        os.remove(path)
        os.rmdir(d)
    return dict(value=round(n * BATCH / dt, 2), unit="frames/s", ms_per_step=round(dt / n * 1e3, 4), dtype="f32",
                note="same workload on the -p 32 engine (fp32 storage, exact-fp32 MFMA): scores within 1e-3 of the CPU "
                     "detector (measured 1e-5); the headline value is the -p 16 engine (fp16, measured 2.9e-3)")


def cpu_baseline(weights, frames, budget_s=12.0):
    """Oracle detector (kind 'port') on this host's cores over a bounded sample of the same frames."""
    import torch
    from oracle.detect import OracleObjectDetector
    from watsor_amd.share import DetectionArray
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("WZ_CPU_BASELINE_THREADS", "64")))
    torch.set_num_threads(cores)
    det = OracleObjectDetector(weights=weights)
    rows = DetectionArray()
    det.detect(frames[0].shape, frames[0], rows)           # warm-up (thread pools, allocations)
    t0 = time.perf_counter()
    done = 0
    lat = []
    while True:
        f = frames[done % len(frames)]
        t1 = time.perf_counter()
        det.detect(f.shape, f, rows)
        lat.append((time.perf_counter() - t1) * 1e3)
        done += 1
        if time.perf_counter() - t0 >= budget_s and done >= 4:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(done / dt, 3), unit="frames/s", cores=cores, kind="port",
                p50_ms=round(float(np.median(lat)), 2),
                sample="%d synthetic %dx%d frames, oracle (torch-CPU fp32 restatement of the reference TF detector), "
                       "%.1f s" % (done, WIDTH, HEIGHT, dt))


class _StubEngine:
    """GPU-less stand-in used only by --dry-run: same surface as HipEngine, a step is a 2 ms sleep."""
    num_slots = 2
    input_size = 300
    device_name = "stub"

    def __init__(self, *a, **k):
        self._busy = {}

    def upload(self, arr):
        return 1

    def submit_device(self, slot, d, ws, hs, cams=None):
        self.wait(slot)
        self._busy[slot] = time.perf_counter() + 0.002

    def wait(self, slot):
        t = self._busy.get(slot, 0) - time.perf_counter()
        if t > 0:
            time.sleep(t)

    def sync(self):
        for s in list(self._busy):
            self.wait(s)

    def slot_rows(self, slot, n):
        return np.zeros((n, 100), dtype=[("confidence", "<f8")])

    def close(self):
        pass


T_START = time.perf_counter()


def note(msg):
    if os.environ.get("WZ_BENCH_VERBOSE", "1") != "0":
        print("[bench %7.2fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the -p 32 engine leg (profiling runs: one engine's kernels only)")
    ap.add_argument("--table", default=None, help="write the per-kernel roofline table (JSON) here")
    ap.add_argument("--dry-run", action="store_true",
                    help="harness self-test without a GPU: a stub engine that sleeps 2 ms per step (used by the "
                         "world_size-2 gloo test; its output is marked invalid)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # libwatsor_hip.so owns the GPU side (its own HIP stream, events, graphs): load it FIRST so it binds
    # to /opt/rocm's HIP runtime, and never initialise torch's bundled copy of that runtime in this
    # process.  torch is used for its CPU side only: the rendezvous/barrier/max-over-ranks (gloo -- the
    # path has no data-path collective) and the oracle's conv2d in the cpu_baseline leg.
    from watsor_amd import engine as builder
    from watsor_amd.synth import synthetic_frame, synthetic_weights
    if args.dry_run:
        HipEngine = _StubEngine
    else:
        from watsor_amd.runtime import HipEngine
        from watsor_amd import _lib
        _lib.load()
        note("library loaded")

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        note("process group up (gloo)")

    model_dir = "/tmp/wz_bench_%d_%d" % (os.getpid(), rank)
    os.makedirs(model_dir, exist_ok=True)
    engine_path = os.path.join(model_dir, "mi355x.bin")
    if args.dry_run:
        weights = None
        open(engine_path, "wb").close()
    else:
        weights = synthetic_weights(1234)
        builder.save_engine(builder.build_engine(weights), engine_path)

    note("engine file built")
    eng = HipEngine(engine_path, local_rank, BATCH, WIDTH, HEIGHT)
    note("engine created on " + eng.device_name)
    # camera `rank`: a ring of 4 batches of distinct frames, pre-staged in HBM
    ring = 4
    host_frames = [synthetic_frame(WIDTH, HEIGHT, 1234 + rank * 1000 + i) for i in range(ring * BATCH if not args.dry_run else 2)]
    if args.dry_run:
        host_frames = (host_frames * (ring * BATCH))[:ring * BATCH]
    d_frames = [eng.upload(f) for f in host_frames]
    ws, hs = [WIDTH] * BATCH, [HEIGHT] * BATCH

    lanes = eng.num_slots

    def submit(step):
        b = step % ring
        eng.submit_device(step % lanes, d_frames[b * BATCH:(b + 1) * BATCH], ws, hs)

    def barrier():
        # barrier + device synchronize on both sides of the timed region (eng.sync() = hipStreamSynchronize
        # of the only stream this process ever enqueues GPU work on)
        eng.sync()
        if dist is not None:
            dist.barrier()

    note("frames staged in HBM")
    for s in range(args.warmup):
        submit(s)
    barrier()
    note("warm-up done")
    t0 = time.perf_counter()
    for s in range(args.steps):
        submit(s)
    barrier()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.3f s" % elapsed)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-step latency (synchronous steps), outside the timed region
    lat = []
    for s in range(min(args.steps, 50)):
        t1 = time.perf_counter()
        submit(s)
        eng.wait(s % lanes)
        lat.append((time.perf_counter() - t1) * 1e3)
    rows = eng.slot_rows((min(args.steps, 50) - 1) % lanes, BATCH)
    detections_per_frame = float((rows["confidence"] > 0).sum()) / BATCH

    out = None
    if rank == 0 and args.dry_run:
        out = {"metric": "DRY RUN (stub engine, no GPU) -- invalid as a measurement",
               "value": round(args.steps * BATCH * world / elapsed, 2), "unit": "frames/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
               "data": "synthetic", "config": {"workload": "dry-run"},
               "camera_seeds": [1234 + r * 1000 for r in range(world)]}
    elif rank == 0:
        note("latency loop done")
        stages = eng.profile_device(d_frames[:BATCH], ws, hs, reps=20)
        note("stage profile done")
        roof, table = roofline_from_stages(stages, eng.ops(), BATCH, WIDTH * HEIGHT * 3, eng.input_size)
        if args.table:
            json.dump(dict(stages=stages, kernels=table), open(args.table, "w"), indent=1)
        frames_total = args.steps * BATCH * world
        out = {
            "metric": "detected frames/sec (whole node) + p50 per-frame latency, SSD-MobileNet 300x300",
            "value": round(frames_total / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "p50_ms": round(float(np.median(lat)), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "1 synthetic 640x480 RGB stream per GPU, batch=8 frames, SSD-MobileNet-v2 300x300 "
                                   "(seeded random-init weights), frames resident in HBM, rows copied back to host",
                       "batch": BATCH, "frame": "%dx%d" % (WIDTH, HEIGHT), "parallelism": "replica-per-gpu x%d" % world, "batches_in_flight": lanes,
                       "detections_per_frame": detections_per_frame},
            "roofline": roof,
        }
        if world == 1:
            eng.close()
        if world == 1 and not args.no_fp32_leg:
            out["fp32_engine"] = fp32_engine_leg(weights, host_frames, rank)
            note("fp32 engine leg done")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, host_frames[:BATCH])
            note("cpu baseline done")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)
    try:
        os.remove(engine_path)
        os.rmdir(model_dir)
    except OSError:
        pass


if __name__ == "__main__":
    main()
