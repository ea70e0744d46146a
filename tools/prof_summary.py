"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel stats + concurrency of the timeline."""
import sqlite3
import sys


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
