"""Phase timestamps of the LDS-DMA convolution kernel wz_k_conv_lds (first / last workgroup), batch 8.

Run with WZ_LDS_RS=0 (the default kernel is the register-staged one, see tools/rs_probe.py)."""
import os, sys, ctypes as C
os.environ.setdefault("WATSOR_HIP_DEV", "1")   # tools run on the development library (stage entry points, knobs, profiling)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["WZ_MB_DEBUG"] = "1"
os.environ.setdefault("WZ_GRAPH", "0")
ITERS = int(os.environ.get("PROBE_ITERS", "3"))
from watsor_amd import engine as eb, _lib
from watsor_amd.synth import synthetic_frame, synthetic_weights
from watsor_amd.runtime import HipEngine
path = "/tmp/wz_probe/mi355x.bin"; os.makedirs("/tmp/wz_probe", exist_ok=True)
eb.save_engine(eb.build_engine(synthetic_weights(1234)), path)
e = HipEngine(path, 0, 8, 640, 480)
frames = [synthetic_frame(640, 480, 1234 + i) for i in range(8)]
d = [e.upload(f) for f in frames]
for it in range(ITERS):
    e.submit_device(0, d, [640] * 8, [480] * 8); e.wait(0)
ops = e.ops()
out = np.zeros((len(ops), 16), np.uint64)
grp = np.zeros(len(ops), np.int32)
_lib.check(e._lib.wz_debug_mbconv(e._h, C.c_void_p(out.ctypes.data), C.c_void_p(grp.ctypes.data)))
for i, o in enumerate(ops):
    if o["kind"] == 4 or not out[i].any():
        continue
    t = out[i].astype(np.int64)
    f = lambda a, b, base=0: (t[base + b] - t[base + a]) / 100.0 if t[base + b] and t[base + a] else float("nan")
    def wg(base):
        steps, cyc = int(t[base + 6] & 0xffff), int(t[base + 6] >> 16)
        loop_us = f(2, 3, base)
        mhz = cyc / max(loop_us, 1e-9)
        return ("%2d steps: prologue %.2f first-data %.2f loop %.2f (%.0f MHz; per step %.0f cyc = wait %.0f + stage %.0f + compute %.0f) store %.2f"
                % (steps, f(0, 1, base), f(1, 2, base), loop_us, mhz, cyc / steps, t[base + 5] / steps, t[base + 7] / steps,
                   (cyc - t[base + 5] - t[base + 7]) / steps, f(3, 4, base)))
    print("%-32s M %5d N %4d K %5d | first WG %s total %.2f | last WG starts +%.2f %s ends +%.2f"
          % (o["name"].split("/")[-1][:32], 8 * o["hout"] * o["wout"], o["cout"], o["ksize"] ** 2 * o["cin"],
             wg(0), f(0, 4), (t[8] - t[0]) / 100.0, wg(8), (t[12] - t[0]) / 100.0))
