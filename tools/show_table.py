"""Print a bench.py --table JSON (per-kernel roofline table + per-stage times)."""
import json
import sys

d = json.load(open(sys.argv[1]))
for k in d["kernels"]:
    print("%-22s n=%3d  %8.1f us/step  avg %7.2f us  %8.1f GB/s %8.1f TF  frac %.3f" % (
        k["kernel"], k["launches"], k["ms_per_step"] * 1e3, k["avg_us"], k["gbs"], k["tflops"], k["t_roof_frac"]))
print("sum %.1f us" % sum(k["ms_per_step"] * 1e3 for k in d["kernels"]))
if len(sys.argv) > 2:
    _c = sorted(ms for n, ms in d["stages"] if n == "(empty)" or n.endswith("#splitk_reduce"))
    _near = [ms for ms in _c if ms <= _c[0] + 1e-3]
    ov = _near[len(_near) // 2]   # median of the empty brackets (see bench.py: empty_bracket_ms)
    for name, ms in d["stages"]:
        v = (ms - ov) * 1e3
        if v > 0.3:
            print("%-70s %8.2f" % (name, v))
