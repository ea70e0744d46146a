"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel stats + concurrency of the timeline."""
import sqlite3
import sys


def launch_shapes(cur):
    """Per kernel: workgroups per launch, threads, registers, LDS -> how many workgroups a CU holds and how many ROUNDS of workgroups a
    launch is on the 256 CUs (a launch whose workgroups each wait out one memory latency takes that many latencies).  Columns are looked
    up by name: the rocpd schema differs between rocprofv3 releases; nothing is printed when they are not there."""
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    except Exception:
        return []

    def pick(*names):
        return next((n for n in names if n in cols), None)
    gx, wx = pick("grid_size_x", "grid_x", "grid_size"), pick("workgroup_size_x", "workgroup_x", "workgroup_size")
    gy, gz = pick("grid_size_y", "grid_y"), pick("grid_size_z", "grid_z")
    wy, wz = pick("workgroup_size_y", "workgroup_y"), pick("workgroup_size_z", "workgroup_z")
    vg, ag = pick("arch_vgpr_count", "vgpr_count"), pick("accum_vgpr_count")
    lds = pick("lds_block_size", "lds_size", "group_segment_size")
    if not (gx and wx and vg):
        return ["", "launch shapes: columns not found in this trace (%s)" % ", ".join(cols)]
    one = lambda c: c if c else "1"
    q = ("select name, %s*%s*%s, %s*%s*%s, %s, %s, %s, count(*), avg(end-start) from kernels group by 1,2,3,4,5,6 order by 8*7 desc"
         % (gx, one(gy), one(gz), wx, one(wy), one(wz), vg, ag or "0", lds or "0"))
    out = ["", "launch shapes (work-items are what the trace calls grid size; rounds = workgroups / (256 CUs x workgroups a CU holds by registers, LDS and wave slots)):",
           "%-92s %7s %5s %5s %7s %7s %7s %7s %8s" % ("kernel", "wgs", "thr", "vgpr", "lds_KiB", "wg/CU", "rounds", "calls", "avg_us")]
    for name, items, thr, v, a, l, calls, avg in cur.execute(q):
        thr = max(int(thr), 1)
        wgs = int(items) // thr
        waves = (thr + 63) // 64
        regs = 2 * max(int(v) + int(a), 8)    # (this rocprofv3 reports HALF of a wave64 kernel's allocation: 88 for a kernel compiled to 173 registers,
                                              #  64 for one at 128, 120 for 237 -- checked against .amdhsa_next_free_vgpr of the same builds)
        by_regs = (512 // regs) * 4 // waves if regs <= 512 else 0      # waves per SIMD by the unified register file, x 4 SIMDs
        by_slots = 32 // waves                                            # 8 wave slots per SIMD
        by_lds = (160 * 1024) // int(l) if l else 99
        per_cu = max(min(by_regs, by_slots, by_lds), 1)
        out.append("%-92s %7d %5d %5d %7.1f %7d %7.2f %7d %8.2f" % (name.replace("void ", "")[:92], wgs, thr, regs, (l or 0) / 1024.0, per_cu,
                                                                   wgs / (256.0 * per_cu), calls, avg / 1e3))
    return out


def main(path, out=None, js=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    lines = ["%-100s %8s %14s %10s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for r in rows:
        lines.append("%-100s %8d %14.1f %10.3f %8.2f" % r)
    ev = list(cur.execute("select start,end,queue_id from kernels order by start"))
    if ev:
        t0, t1 = ev[0][0], max(e[1] for e in ev)
        busy = sum(e[1] - e[0] for e in ev)
        # union of busy intervals
        union, cs, ce = 0, ev[0][0], ev[0][1]
        for s, e, _ in ev[1:]:
            if s > ce:
                union += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        union += ce - cs
        queues = sorted(set(e[2] for e in ev))
        lines.append("")
        lines.append("timeline: %d dispatches on %d queues over %.3f ms; sum of kernel durations %.3f ms; "
                     "union (GPU busy with >=1 kernel) %.3f ms; mean concurrency %.2f"
                     % (len(ev), len(queues), (t1 - t0) / 1e6, busy / 1e6, union / 1e6, busy / max(union, 1)))
    lines += launch_shapes(cur)
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "a").write(text + "\n")
    if js:   # per bench.py kernel class: mean per-dispatch duration (read back by bench.py's roofline object)
        import json
        agg = {}
        for name, calls, _, avg, _ in rows:
            k = name.replace("void ", "").split("(")[0].split("<")[0]
            if k.startswith("_Z"):
                k = "wz_k_stem" if "stem" in k else "wz_k_preprocess" if "preprocess" in k else k
            if k.startswith("wz_k_mbconv_hp"):
                k = "wz_k_mbconv_hp"
            elif k.startswith("wz_k_mbconv"):
                k = "wz_k_mbconv"
            if k in ("wz_k_conv_lds", "wz_k_conv"):
                k = "wz_k_conv<%s>" % name.split("<")[1].split(",")[0].split(">")[0]
            a = agg.setdefault(k, [0.0, 0])
            a[0] += avg * calls                                # the `average` column is what the table prints as avg_us
            a[1] += calls
        outd = {k: dict(avg_us=round(v[0] / v[1], 3), calls=v[1]) for k, v in agg.items()}
        outd["_source"] = "rocprofv3 --kernel-trace --stats of bench.py (tools/prof_summary.py)"
        json.dump(outd, open(js, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
