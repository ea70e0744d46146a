"""Which kernels of DIFFERENT lanes are on the chip at the same time -- measured without a profiler.

rocprofv3's kernel trace serialises the four lanes of this engine (concurrency 1.18 in profiles/r04zz_*_lanes4: 24 k frames/s under it),
so the committed traces cannot show what the headline number rests on: two and more batches' kernels co-resident.  This tool runs the
headline workload (BASELINE configs[1]: batches of 8 frames 640x480 resident in HBM, submitted round-robin onto the lanes) on
`libwatsor_hip_stamps.so` (`make -C watsor_amd/csrc stamps`: the development library with -DWZ_LANE_STAMPS=1), in which every kernel
records when its first workgroup entered and its last one left, on the device's constant 100 MHz clock (one clock for all XCDs).  From
the (lane, step, launch) intervals of an UNPROFILED run it prints

  * frames/s of this very run (the stamps cost ~2 %: compare with the product library's figure printed beside it),
  * kernels in flight: mean over the time the chip is busy, and the share of time at 0, 1, 2, 3, 4+,
  * per launch of a batch: kernel, workgroups, threads, registers, LDS, workgroups a CU holds (the runtime's occupancy calculator),
    mean duration under load, and its CU-SLOT-TIME: duration x the share of the chip's workgroup slots the launch books,
    min(1, workgroups / (256 CUs x workgroups per CU)) -- what a launch takes away from the other lanes,
  * the slot-time sum per step against the measured ms per step.

    python tools/lane_overlap.py [--steps 400] [--default-program] [--out profiles/r05_lane_overlap.txt] [--json profiles/lane_overlap.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_LIB = os.path.join(ROOT, "watsor_amd", "libwatsor_hip_stamps.so")
NUM_CUS = 256
TICK_US = 0.01            # s_memrealtime: 100 MHz


def collect(engine_path, steps, warm, batch):
    """-> (intervals [(lane, step, launch, t0_ticks, t1_ticks)], launches [dict], frames/s, ms per step, lanes)"""
    from watsor_amd.runtime import HipEngine
    from watsor_amd.synth import synthetic_frame
    eng = HipEngine(engine_path, 0, batch, 640, 480, dev=True)
    lib = eng._lib
    ring = 5
    d = [eng.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(ring * batch)]
    ws, hs = [640] * batch, [480] * batch
    lanes = eng.num_slots
    buf = (C.c_uint64 * (2 * 160))()
    pending = {}
    out = []

    def retire(lane):
        eng.wait(lane)
        k = lib.wz_debug_lane_stamps(eng._h, lane, buf, 160)
        if k <= 0:
            raise RuntimeError("no stamps: " + lib.wz_last_error().decode())
        out.append((lane, pending.pop(lane), C.string_at(buf, 16 * k)))      # (parsed after the run: the loop stays as light as the bench's)

    def submit(step):
        lane = step % lanes
        if lane in pending:
            retire(lane)
        b = step % ring
        eng.submit_device(lane, d[b * batch:(b + 1) * batch], ws, hs)
        pending[lane] = step

    for s in range(warm):
        submit(s)
    for lane in list(pending):
        retire(lane)
    out.clear()
    eng.sync()
    t0 = time.perf_counter()
    for s in range(steps):
        submit(s)
    for lane in sorted(pending, key=lambda l: pending[l]):
        retire(lane)
    eng.sync()
    dt = time.perf_counter() - t0
    # what was launched (the last batch's graph: the same for every step)
    name = C.create_string_buffer(512)
    dims = (C.c_int32 * 8)()
    launches = []
    n = lib.wz_debug_lane_launch(eng._h, 0, 0, name, 512, dims)
    for i in range(max(n, 0)):
        lib.wz_debug_lane_launch(eng._h, 0, i, name, 512, dims)
        launches.append(dict(kernel=name.value.decode(), workgroups=dims[0], threads=dims[1], lds_bytes=dims[2], wg_per_cu=dims[3], registers=dims[4]))
    eng.close()
    import struct
    iv = []
    for lane, step, raw in out:
        w = struct.unpack("<%dQ" % (len(raw) // 8), raw)
        iv.extend((lane, step, i, w[2 * i], w[2 * i + 1]) for i in range(len(w) // 2))
    return iv, launches, steps * batch / dt, dt / steps * 1e3, lanes


def product_rate(engine_path, steps, warm, batch):
    """The same loop on the PRODUCT library (no stamps, no waits in between): what the stamps cost."""
    from watsor_amd.runtime import HipEngine
    from watsor_amd.synth import synthetic_frame
    eng = HipEngine(engine_path, 0, batch, 640, 480, dev=False)
    d = [eng.upload(synthetic_frame(640, 480, 1234 + i)) for i in range(5 * batch)]
    ws, hs = [640] * batch, [480] * batch
    lanes = eng.num_slots
    for s in range(warm):
        eng.submit_device(s % lanes, d[(s % 5) * batch:(s % 5 + 1) * batch], ws, hs)
    eng.sync()
    t0 = time.perf_counter()
    for s in range(steps):
        eng.submit_device(s % lanes, d[(s % 5) * batch:(s % 5 + 1) * batch], ws, hs)
    eng.sync()
    dt = time.perf_counter() - t0
    eng.close()
    return steps * batch / dt


def short(name):
    name = name.replace("void ", "")
    return name if len(name) <= 58 else name[:55] + "..."


def analyse(iv, launches, fps, ms_step, lanes, steps, batch, fps_product):
    bad = [x for x in iv if x[4] <= x[3] or x[3] == 0 or x[3] >= (1 << 63)]
    iv = [x for x in iv if not (x[4] <= x[3] or x[3] == 0 or x[3] >= (1 << 63))]
    # steady state: leave out the first and the last `lanes` steps (the pipeline fills and drains there)
    lo, hi = lanes, steps - lanes
    mid = [x for x in iv if lo <= x[1] < hi]
    t_begin = min(x[3] for x in mid)
    t_end = max(x[4] for x in mid)
    # every interval that overlaps the window, clipped to it (kernels of the excluded steps that run inside it DO hold the chip)
    clip = [(max(x[3], t_begin), min(x[4], t_end)) for x in iv if x[4] > t_begin and x[3] < t_end]
    ev = []
    for a, b in clip:
        ev.append((a, 1))
        ev.append((b, -1))
    ev.sort()
    at = {}
    depth, last = 0, t_begin
    for t, dlt in ev:
        at[depth] = at.get(depth, 0) + (t - last)
        last = t
        depth += dlt
    window = t_end - t_begin
    busy = window - at.get(0, 0)
    total = sum(b - a for a, b in clip)
    n_launch = max(x[2] for x in iv) + 1
    per = []
    for i in range(n_launch):
        durs = [(x[4] - x[3]) * TICK_US for x in mid if x[2] == i]
        info = launches[i] if i < len(launches) else dict(kernel="?", workgroups=0, threads=0, lds_bytes=0, wg_per_cu=0, registers=0)
        cap = NUM_CUS * max(info["wg_per_cu"], 1)
        share = min(1.0, info["workgroups"] / cap) if info["workgroups"] else 0.0
        mean = sum(durs) / max(len(durs), 1)
        per.append(dict(launch=i, **info, mean_us=round(mean, 2), rounds=round(info["workgroups"] / cap, 3) if info["workgroups"] else 0.0,
                        chip_share=round(share, 3), slot_time_us=round(mean * share, 2),
                        uncapped_us=round(mean * info["workgroups"] / cap, 2) if info["workgroups"] else 0.0))
    steps_in_window = hi - lo
    res = dict(
        frames_per_s_stamped_run=round(fps, 1), ms_per_step_stamped_run=round(ms_step, 4), frames_per_s_product_library=round(fps_product, 1) if fps_product else None,
        lanes=lanes, steps=steps, batch=batch, launches_per_step=n_launch, intervals=len(iv), intervals_dropped=len(bad),
        window_us=round(window * TICK_US, 1), us_per_step_device_clock=round(window * TICK_US / steps_in_window, 2),
        kernels_in_flight_mean_while_busy=round(total / busy, 3), kernels_in_flight_mean_over_window=round(total / window, 3),
        chip_idle_share=round(at.get(0, 0) / window, 4),
        time_share_by_kernels_in_flight={str(k): round(v / window, 4) for k, v in sorted(at.items())},
        kernel_time_sum_us_per_step=round(sum(p["mean_us"] for p in per), 1),
        cu_slot_time_us_per_step=round(sum(p["slot_time_us"] for p in per), 1),
        cu_slot_time_uncapped_us_per_step=round(sum(p["uncapped_us"] for p in per), 1),
        per_launch=per)
    res["slot_time_over_step_time"] = round(res["cu_slot_time_us_per_step"] / res["us_per_step_device_clock"], 3)
    return res


def render(res, program):
    L = []
    L.append("lane overlap of the headline workload, %s program -- tools/lane_overlap.py on libwatsor_hip_stamps.so (in-kernel entry / exit stamps, no profiler)" % program)
    L.append("run: %d steps of batch %d over %d lanes: %.0f frames/s, %.4f ms per step (host clock); the product library on the same loop: %s frames/s"
             % (res["steps"], res["batch"], res["lanes"], res["frames_per_s_stamped_run"], res["ms_per_step_stamped_run"],
                "%.0f" % res["frames_per_s_product_library"] if res["frames_per_s_product_library"] else "n/a"))
    L.append("steady-state window (first and last %d steps left out): %.1f us on the device clock = %.2f us per step; %d launches per step, %d intervals (%d dropped)"
             % (res["lanes"], res["window_us"], res["us_per_step_device_clock"], res["launches_per_step"], res["intervals"], res["intervals_dropped"]))
    L.append("")
    L.append("KERNELS IN FLIGHT: mean %.2f while the chip is busy (%.2f over the whole window; chip idle %.1f %% of it)"
             % (res["kernels_in_flight_mean_while_busy"], res["kernels_in_flight_mean_over_window"], 100 * res["chip_idle_share"]))
    L.append("  share of the window with k kernels in flight: " + "  ".join("%s: %.1f %%" % (k, 100 * v) for k, v in res["time_share_by_kernels_in_flight"].items()))
    L.append("")
    L.append("per launch of a batch (duration = first workgroup's entry to last workgroup's exit, mean over the window, WITH the other lanes' kernels on the chip):")
    L.append("%3s %-58s %6s %5s %5s %7s %6s %7s %8s %6s %9s" % ("#", "kernel", "wgs", "thr", "regs", "lds_KiB", "wg/CU", "rounds", "mean_us", "share", "slot_us"))
    for p in res["per_launch"]:
        L.append("%3d %-58s %6d %5d %5d %7.1f %6d %7.2f %8.2f %6.2f %9.2f" % (p["launch"], short(p["kernel"]), p["workgroups"], p["threads"], p["registers"],
                                                                           p["lds_bytes"] / 1024.0, p["wg_per_cu"], p["rounds"], p["mean_us"], p["chip_share"], p["slot_time_us"]))
    L.append("")
    L.append("sum of the launches' durations per step: %.1f us (= %.2f x the step time: that many kernels are in flight on average)"
             % (res["kernel_time_sum_us_per_step"], res["kernel_time_sum_us_per_step"] / res["us_per_step_device_clock"]))
    L.append("CU-SLOT-TIME per step: %.1f us (share of the chip's workgroup slots a launch books x its duration, summed; uncapped workgroups / slots x duration: %.1f us)"
             % (res["cu_slot_time_us_per_step"], res["cu_slot_time_uncapped_us_per_step"]))
    L.append("step time on the device clock: %.2f us  ->  slot-time / step time = %.2f" % (res["us_per_step_device_clock"], res["slot_time_over_step_time"]))
    top = sorted(res["per_launch"], key=lambda p: -p["slot_time_us"])[:8]
    L.append("largest holders: " + ", ".join("#%d %.1f us" % (p["launch"], p["slot_time_us"]) for p in top))
    return "\n".join(L)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warm", type=int, default=40)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--default-program", action="store_true", help="the default -p 16 program instead of the robust one (the headline's)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-product", action="store_true")
    args = ap.parse_args()
    if not os.path.isfile(STAMPS_LIB):
        print("missing %s: make -C watsor_amd/csrc stamps" % STAMPS_LIB, file=sys.stderr)
        return 2
    os.environ["WATSOR_HIP_DEV_LIBRARY"] = STAMPS_LIB       # read when watsor_amd._lib is imported
    os.environ.setdefault("WZ_GRAPH", "1")                  # what was launched is noted when the graph is captured: a batch that finds the other lanes idle
                                                            # (every batch of a one-lane run) would otherwise go kernel by kernel and leave no notes
    from watsor_amd import engine as eb
    from watsor_amd.synth import synthetic_weights
    path = "/tmp/wz_lane_overlap_%d/mi355x.bin" % os.getpid()
    eb.save_engine(eb.build_engine(synthetic_weights(1234), robust=not args.default_program), path)
    fps_product = None if args.no_product else product_rate(path, args.steps, args.warm, args.batch)
    iv, launches, fps, ms_step, lanes = collect(path, args.steps, args.warm, args.batch)
    os.remove(path)
    res = analyse(iv, launches, fps, ms_step, lanes, args.steps, args.batch, fps_product)
    program = "default" if args.default_program else "robust"
    res["program"] = program
    text = render(res, program)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
