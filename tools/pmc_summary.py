"""Summarise rocprofv3 --pmc runs (rocpd sqlite): per kernel, mean counter value per dispatch.

    python tools/pmc_summary.py <results.db> [<results.db> ...]  [--out file]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  On gfx950 FETCH_SIZE counts a
128-byte request of a wide (16 B/lane) read as 64 bytes (MI355X_MICROARCH.md, HBM section): the column
`fetch_corrected` doubles it.  WRITE_SIZE is uncalibrated there and is reported as is.
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for pre in ("void ",):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:60]


def main(paths, out=None):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, n]
    dur = defaultdict(lambda: [0.0, 0])
    for p in paths:
        db = sqlite3.connect(p)
        for kname, cname, value, d in db.execute(
                "select kernel_name, counter_name, value, duration from counters_collection"):
            a = agg[short(kname)][cname]
            a[0] += value
            a[1] += 1
            dd = dur[short(kname)]
            dd[0] += d
            dd[1] += 1
    counters = sorted({c for k in agg for c in agg[k]})
    lines = ["%-62s %8s %10s " % ("kernel", "launches", "avg_us") + " ".join("%22s" % c for c in counters)]
    for k in sorted(agg, key=lambda k: -dur[k][0]):
        n = max(a[1] for a in agg[k].values())
        row = "%-62s %8d %10.2f " % (k, n, dur[k][0] / max(dur[k][1], 1) / 1e3)
        row += " ".join("%22.1f" % (agg[k][c][0] / agg[k][c][1]) if c in agg[k] else "%22s" % "-" for c in counters)
        lines.append(row)
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")
    return agg, dur


def traffic_json(agg, path):
    """Per kernel CLASS (template arguments stripped, the way bench.py names kernels): mean HBM-side bytes per
    launch.  fetch_bytes_corrected doubles FETCH_SIZE (gfx950 counts a 128-B request of a 16 B/lane read as 64 B)."""
    import json
    cls = defaultdict(lambda: dict(fetch_kib=0.0, write_kib=0.0, nf=0, nw=0))
    for k, counters in agg.items():
        c = k.split("<")[0]
        if c.startswith("_Z"):
            c = "wz_k_stem" if "stem" in c else "wz_k_preprocess" if "preprocess" in c else c
        if c.startswith("wz_k_mbconv_hp"):
            c = "wz_k_mbconv_hp"                  # the split-operand blocks (0 .. 12 of the default -p 16 program)
        elif c.startswith("wz_k_mbconv"):
            c = "wz_k_mbconv"                     # the plain fused-block kernels serve one op class
        if c == "wz_k_conv_lds" or c == "wz_k_conv":
            c = "wz_k_conv<%s>" % k.split("<")[1].split(",")[0].split(">")[0]
        if "FETCH_SIZE" in counters:
            cls[c]["fetch_kib"] += counters["FETCH_SIZE"][0]; cls[c]["nf"] += counters["FETCH_SIZE"][1]
        if "WRITE_SIZE" in counters:
            cls[c]["write_kib"] += counters["WRITE_SIZE"][0]; cls[c]["nw"] += counters["WRITE_SIZE"][1]
    outd = {}
    for c, v in cls.items():
        f = v["fetch_kib"] / v["nf"] * 1024 if v["nf"] else None
        w = v["write_kib"] / v["nw"] * 1024 if v["nw"] else None
        outd[c] = dict(fetch_bytes_raw=f, fetch_bytes_corrected=2 * f if f is not None else None, write_bytes=w,
                       launches_sampled=v["nf"])
    json.dump(outd, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    out = js = None
    if "--json" in args:
        i = args.index("--json")
        js = args[i + 1]
        del args[i:i + 2]
    if "--out" in args:
        i = args.index("--out")
        out = args[i + 1]
        del args[i:i + 2]
    agg, dur = main(args, out)
    if js:
        traffic_json(agg, js)
