"""Phase timestamps of the fused inverted-residual block kernels (first / last workgroup), batch 8."""
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
    if o["kind"] != 4:
        continue
    t = out[i].astype(np.int64)
    f = lambda a, b, base=0: (t[base + b] - t[base + a]) / 100.0 if t[base + b] and t[base + a] else float("nan")
    print("%-16s %3dx%-3d cin %3d cmid %3d cout %3d s%d groups %2d | first WG: loads %.2f expand %.2f dw+proj %.2f store %.2f total %.2f | last WG starts +%.2f, total %.2f, ends +%.2f | %.0f MHz"
          % (o["name"].split("/")[-1], o["hin"], o["win"], o["cin"], o["cmid"], o["cout"], o["stride"], grp[i],
             f(0, 1), f(1, 2), f(2, 3), f(3, 4), f(0, 4), (t[8] - t[0]) / 100.0, f(0, 4, 8), (t[12] - t[0]) / 100.0,
             (t[6] - t[5]) / max(t[4] - t[0], 1) * 100.0))
